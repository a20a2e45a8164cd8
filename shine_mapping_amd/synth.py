"""Synthetic LiDAR workloads shaped like the reference's datasets (there is no network for the real ones).

SURVEY.md §8(d): an analytic street canyon ray-cast from a 64-beam spinning sensor, then the reference's
per-ray sampling scheme (utils/data_sampler.py:18-139: uniform +-range around the hit, uniform in free
space from 0.3 x range to a bit behind the surface), giving the (coord, sdf_label, weight) sample pool
that LiDARDataset.get_batch draws from (dataset/lidar_dataset.py:430-450).  Everything is torch and
device-agnostic, so bench.py generates the pool on the GPU in a second or two.

Workload presets follow the shipped yamls:
  maicity : config/maicity/maicity_batch.yaml  leaf 0.2 m, L=3 (or 4), sigma 0.05, BCE only
  kitti   : config/kitti/kitti_batch.yaml      leaf 0.3 m, L=3, sigma 0.1, eikonal on (w_e 0.1); 600 m polyline with
            two turns
  kitti_large : the same sensor/config on an 8 x 1 km serpentine (8.4 km of trajectory): a map whose feature tables
            (> 256 MiB) do not fit the Infinity Cache — the regime dataset/lidar_dataset.py:93-97 anticipates
  ncd     : config/ncd/ncd_incre_reg.yaml      leaf 0.2 m, L=3, sigma 0.1, sum reduction + regulariser
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

PRESETS = {
    "maicity": dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.2, sigma_sigmoid_m=0.05,
                    surface_sample_range_m=0.15, surface_sample_n=3, free_sample_n=3, free_sample_begin_ratio=0.3,
                    free_sample_end_dist_m=0.8, ekional_loss_on=False, weight_e=0.1, loss_reduction="mean",
                    pc_radius_m=50.0, min_range_m=1.5, street_len=100.0, turns=0, lr=0.01),
    "kitti": dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.3, sigma_sigmoid_m=0.1,
                  surface_sample_range_m=0.3, surface_sample_n=3, free_sample_n=3, free_sample_begin_ratio=0.3,
                  free_sample_end_dist_m=0.8, ekional_loss_on=True, weight_e=0.1, loss_reduction="mean",
                  pc_radius_m=50.0, min_range_m=3.0, street_len=600.0, turns=2, lr=0.01),
    "kitti_large": dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.3, sigma_sigmoid_m=0.1,
                        surface_sample_range_m=0.3, surface_sample_n=3, free_sample_n=3, free_sample_begin_ratio=0.3,
                        free_sample_end_dist_m=0.8, ekional_loss_on=True, weight_e=0.1, loss_reduction="mean",
                        pc_radius_m=50.0, min_range_m=3.0, street_len=8420.0, turns=-8, lr=0.01),
    "ncd": dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.2, sigma_sigmoid_m=0.1,
                surface_sample_range_m=0.3, surface_sample_n=3, free_sample_n=3, free_sample_begin_ratio=0.3,
                free_sample_end_dist_m=1.0, ekional_loss_on=False, weight_e=0.1, loss_reduction="sum",
                lambda_forget=1e4, pc_radius_m=25.0, min_range_m=1.5, street_len=40.0, turns=0, lr=0.01),
}


def make_config(kind: str, device="cuda", **over) -> SimpleNamespace:
    """Attribute bag with the names utils/config.py uses (so FeatureOctree / Decoder take it as-is)."""
    c = SimpleNamespace(
        feature_dim=8, feature_std=0.05, poly_int_on=True, geo_mlp_level=2, geo_mlp_hidden_dim=32,
        geo_mlp_bias_on=True, sem_mlp_level=2, sem_mlp_hidden_dim=32, sem_mlp_bias_on=True, sem_class_count=20,
        logistic_gaussian_ratio=0.55, lambda_forget=0.0, device=device, bs=4096, weight_decay=1e-7,
        mc_vis_level=1, pad_voxel=2, dtype=torch.float32, time_conditioned=False,  # utils/config.py:140-146,27
    )
    c.__dict__.update(PRESETS[kind])
    c.__dict__.update(over)
    c.scale = 1.0 / (c.leaf_vox_size * (2 ** (c.tree_level_world - 1)))  # utils/config.py:372-374
    c.sigma_sigmoid = c.logistic_gaussian_ratio * c.sigma_sigmoid_m * c.scale  # shine_batch.py:87
    return c


def _boxes(street_len: float, gen: torch.Generator, n=20):
    cx = torch.rand(n, generator=gen) * street_len
    cy = (torch.rand(n, generator=gen) * 2 - 1) * 6.0
    sx = torch.rand(n, generator=gen) * 3 + 1
    sy = torch.rand(n, generator=gen) * 2 + 1
    sz = torch.rand(n, generator=gen) * 2.5 + 0.5
    lo = torch.stack((cx - sx / 2, cy - sy / 2, torch.zeros(n)), 1)
    hi = torch.stack((cx + sx / 2, cy + sy / 2, sz), 1)
    return lo, hi


def cast_scan(origin, dirs, lo, hi, half_width=8.0, facade_h=10.0, max_range=50.0, min_range=1.5):
    """Analytic ray cast of one scan: ground z=0, facades y=+-half_width, axis-aligned boxes. -> hit points [M,3]."""
    o = origin
    big = torch.full((dirs.shape[0],), float("inf"), device=dirs.device)
    dz = dirs[:, 2]
    t = torch.where(dz < -1e-6, -o[2] / dz, big)
    for sgn in (-1.0, 1.0):
        dy = dirs[:, 1]
        tf = torch.where(dy * sgn > 1e-6, (sgn * half_width - o[1]) / dy, big)
        zf = o[2] + tf * dz
        tf = torch.where((zf >= 0) & (zf <= facade_h), tf, big)
        t = torch.minimum(t, tf)
    inv = 1.0 / torch.where(dirs.abs() < 1e-9, torch.full_like(dirs, 1e-9), dirs)
    t0 = (lo[None] - o[None, None]) * inv[:, None]
    t1 = (hi[None] - o[None, None]) * inv[:, None]
    tn = torch.minimum(t0, t1).amax(-1)
    tx = torch.maximum(t0, t1).amin(-1)
    tb = torch.where((tx >= tn) & (tn > 0), tn, torch.full_like(tn, float("inf"))).amin(-1)
    t = torch.minimum(t, tb)
    ok = (t < max_range) & (t > min_range)
    return o[None] + t[ok, None] * dirs[ok]


def sensor_dirs(beams=64, azimuths=450, fov=(-24.8, 2.0), device="cpu"):
    el = torch.deg2rad(torch.linspace(fov[0], fov[1], beams, device=device))
    az = torch.linspace(0, 2 * math.pi, azimuths + 1, device=device)[:-1]
    ce, se = torch.cos(el)[:, None], torch.sin(el)[:, None]
    d = torch.stack((ce * torch.cos(az)[None], ce * torch.sin(az)[None], se.expand(-1, azimuths)), -1)
    return d.reshape(-1, 3)


def sample_rays(points, origin, cfg, gen=None):
    """Per-ray samples in the [-1,1] space: utils/data_sampler.py:18-139 (labels/normals off). Ray-major order."""
    dev = points.device
    ns, nf = cfg.surface_sample_n, cfg.free_sample_n
    S = ns + nf
    rel = points - origin
    m = rel.shape[0]
    dist = torch.linalg.norm(rel, dim=1, keepdim=True)
    r_s = cfg.surface_sample_range_m * cfg.scale
    disp_s = (torch.rand(m * ns, 1, device=dev, generator=gen) - 0.5) * 2 * r_s
    ratio_s = disp_s / dist.repeat(ns, 1) + 1.0
    rd = dist.repeat(nf, 1)
    hi = cfg.free_sample_end_dist_m * cfg.scale / rd + 1.0
    lo = cfg.free_sample_begin_ratio
    ratio_f = torch.rand(m * nf, 1, device=dev, generator=gen) * (hi - lo) + lo
    disp_f = (ratio_f - 1.0) * rd
    disp = torch.cat((disp_s, disp_f), 0)
    ratio = torch.cat((ratio_s, ratio_f), 0)
    xyz = rel.repeat(S, 1) * ratio + origin
    w = torch.ones(m * S, device=dev)
    w[m * ns:] = -1.0
    xyz = xyz.reshape(S, -1, 3).transpose(0, 1).reshape(-1, 3)
    label = disp.squeeze(1).reshape(S, -1).transpose(0, 1).reshape(-1)
    w = w.reshape(S, -1).transpose(0, 1).reshape(-1)
    return xyz.contiguous(), label.contiguous(), w.contiguous()


def trajectory(street_len: float, turns: int):
    """The sensor path as axis-aligned segments [(x0, y0, heading_deg, length)].
    turns == 0: one straight street; turns > 0: a polyline with that many 90-degree turns (alternating left/right, equal
    legs — SURVEY.md §8d's KITTI-like "600 m polyline with two turns"); turns < 0: a serpentine of |turns| long rows
    joined by 60 m connectors (a large map that still fits the [-1,1] cube)."""
    if turns == 0:
        return [(0.0, 0.0, 0.0, float(street_len))]
    segs, x, y = [], 0.0, 0.0
    if turns > 0:
        leg = street_len / (turns + 1)
        for k in range(turns + 1):
            h = 0.0 if k % 2 == 0 else 90.0
            segs.append((x, y, h, leg))
            x, y = (x + leg, y) if h == 0.0 else (x, y + leg)
        return segs
    rows, gap = -turns, 60.0
    row_len = (street_len - gap * (rows - 1)) / rows
    for k in range(rows):
        h = 0.0 if k % 2 == 0 else 180.0
        segs.append((x, y, h, row_len))
        x = x + row_len if h == 0.0 else x - row_len
        if k + 1 < rows:
            segs.append((x, y, 90.0, gap))
            y += gap
    return segs


def _rot(heading_deg: float, device):
    c, s_ = round(math.cos(math.radians(heading_deg))), round(math.sin(math.radians(heading_deg)))
    return torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], device=device)


def make_frames(cfg, frames=100, beams=64, azimuths=450, seed=42, device="cuda"):
    """Yield per-frame (coord, sdf_label, weight) in the scaled space, like LiDARDataset.process_frame (:115-233).
    Each segment of the trajectory is a street canyon in its own frame (ground, two facades, boxes); a scan is ray-cast
    against the canyon of the segment the sensor is in."""
    g = torch.Generator().manual_seed(seed)
    segs = trajectory(cfg.street_len, getattr(cfg, "turns", 0))
    boxes = []
    for (_, _, _, length) in segs:
        lo, hi = _boxes(length, g, n=max(4, int(20 * length / 100.0)) if len(segs) > 1 else 20)
        boxes.append((lo.to(device), hi.to(device)))
    dirs = sensor_dirs(beams, azimuths, device=device)
    gen = torch.Generator(device=device).manual_seed(seed + 1) if str(device) != "cpu" else \
        torch.Generator().manual_seed(seed + 1)
    total = sum(sg[3] for sg in segs)
    step = total / max(frames, 1)
    # centre the map on the origin so it fits the [-1,1] cube (the reference shifts by the first pose, first_frame_ref)
    ends = [(x0 + math.cos(math.radians(h)) * ln, y0 + math.sin(math.radians(h)) * ln) for (x0, y0, h, ln) in segs]
    xs = [sg[0] for sg in segs] + [e[0] for e in ends]
    ys = [sg[1] for sg in segs] + [e[1] for e in ends]
    shift = torch.tensor([(min(xs) + max(xs)) / 2, (min(ys) + max(ys)) / 2, 0.0], device=device)
    k, acc = 0, 0.0
    rots = [(_rot(sg[2], device), _rot(-sg[2], device)) for sg in segs]
    for f in range(frames):
        s_ = f * step
        while k + 1 < len(segs) and s_ >= acc + segs[k][3]:
            acc += segs[k][3]
            k += 1
        x0, y0, _, _ = segs[k]
        R, Rinv = rots[k]
        lo, hi = boxes[k]
        local_origin = torch.tensor([s_ - acc, 0.3 * math.sin(0.2 * f), 1.8], device=device)
        hits = cast_scan(local_origin, dirs @ Rinv.T if len(segs) > 1 else dirs, lo, hi, max_range=cfg.pc_radius_m,
                         min_range=cfg.min_range_m)
        base = torch.tensor([x0, y0, 0.0], device=device)
        if len(segs) > 1:
            hits = hits @ R.T + base
            origin = local_origin @ R.T + base
        else:
            origin = local_origin
        yield sample_rays((hits - shift) * cfg.scale, (origin - shift) * cfg.scale, cfg, gen)


def build_workload(kind="maicity", frames=100, device="cuda", seed=42, beams=64, azimuths=450, **over):
    """Pool + octree + decoder for a preset: what shine_batch.py:69-95 has in hand when the hot loop starts."""
    import hashlib
    import os

    from .decoder import Decoder
    from .feature_octree import FeatureOctree

    cfg = make_config(kind, device=device, **over)
    # measurement aid: SHINE_WORKLOAD_CACHE=<dir> keeps the built workload on disk, so that the profiler passes of
    # tools/collect_profiles.sh (one process per counter group) build a large map once instead of once per pass
    cache = os.environ.get("SHINE_WORKLOAD_CACHE")
    path = None
    if cache:
        key = repr((kind, frames, seed, beams, azimuths, sorted(over.items())))
        path = os.path.join(cache, "workload_%s.pt" % hashlib.sha1(key.encode()).hexdigest()[:16])
        if os.path.isfile(path):
            blob = torch.load(path, weights_only=False)
            octree, decoder = blob["octree"], blob["decoder"].to(device)
            pool = SimpleNamespace(**{k: v.to(device) for k, v in blob["pool"].items()})
            return SimpleNamespace(cfg=cfg, octree=octree, decoder=decoder, pool=pool)
    torch.manual_seed(seed)
    octree = FeatureOctree(cfg)
    decoder = Decoder(cfg)
    coords, labels, weights = [], [], []
    for c, l, w in make_frames(cfg, frames, beams, azimuths, seed, device):
        octree.update(c[w > 0], False)  # octree_from_surface_samples: True (lidar_dataset.py:213-215)
        coords.append(c)
        labels.append(l)
        weights.append(w)
    pool = SimpleNamespace(coord=torch.cat(coords), sdf_label=torch.cat(labels), weight=torch.cat(weights))
    if path is not None:
        os.makedirs(cache, exist_ok=True)
        torch.save(dict(octree=octree, decoder=decoder, pool={k: v.cpu() for k, v in vars(pool).items()}), path)
    return SimpleNamespace(cfg=cfg, octree=octree, decoder=decoder, pool=pool)


def draw_batch(pool, n, gen=None):
    """LiDARDataset.get_batch, point-sample branch (dataset/lidar_dataset.py:430-450)."""
    idx = torch.randint(0, pool.sdf_label.shape[0], (n,), device=pool.sdf_label.device, generator=gen)
    return pool.coord[idx, :], pool.sdf_label[idx], pool.weight[idx]
