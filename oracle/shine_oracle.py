"""CPU oracle for the SHINE SDF training hot path (query -> decode -> loss -> backward).

TEST INFRASTRUCTURE ONLY — the checker, never the product.  Only ``tests/``,
``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import
this file; ``shine_mapping_amd`` must not (its ops fail loudly when the HIP
library is missing instead of falling back to anything in here).

It is a torch-CPU fp32 restatement of the reference's algorithm, function by
function, with the reference file:line each one follows.  The arithmetic is
written with the same torch op sequence as the reference so that, on the same
machine, it is bit-identical to the reference's own Python (pinned by
oracle/make_golden.py against /root/reference, and by the committed fixtures
in tests/golden/ everywhere else).  The kaolin integer ops come from
oracle/kaolin_shim.py (parity unpinned for those five, see its header).

Because it keeps the reference's per-point Python dict lookup
(model/feature_octree.py:209) it is also the honest ``cpu_baseline`` of kind
"port": same algorithmic structure, same torch CPU kernels, same cost profile.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch

from . import kaolin_shim as kal

_MISS = [-1] * 8


def make_config(**kw) -> SimpleNamespace:
    """The subset of utils/config.py:7-204 the hot path reads, with the same defaults."""
    c = SimpleNamespace(
        tree_level_world=10,
        tree_level_feat=4,
        leaf_vox_size=0.5,
        feature_dim=8,
        feature_std=0.05,
        poly_int_on=True,
        geo_mlp_level=2,
        geo_mlp_hidden_dim=32,
        geo_mlp_bias_on=True,
        sem_class_count=20,
        sigma_sigmoid_m=0.1,
        logistic_gaussian_ratio=0.55,
        loss_reduction="mean",
        ekional_loss_on=False,
        weight_e=0.1,
        loss_weight_on=False,
        lambda_forget=1e5,
        surface_sample_range_m=0.5,
        surface_sample_n=5,
        free_sample_begin_ratio=0.3,
        free_sample_end_dist_m=0.5,
        free_sample_n=2,
        scale=1.0,
    )
    c.__dict__.update(kw)
    # utils/config.py:372-374  calculate_world_scale
    c.scale = 1.0 / (c.leaf_vox_size * (2 ** (c.tree_level_world - 1)))
    return c


# --------------------------------------------------------------------------- octree


class OracleOctree:
    """Restates model/feature_octree.py:29-255 (state, update, get_indices, interpolat, query, regulariser)."""

    def __init__(self, cfg):
        self.max_level = cfg.tree_level_world
        self.featured_level_num = cfg.tree_level_feat
        self.free_level_num = self.max_level - self.featured_level_num + 1  # :40
        self.feature_dim = cfg.feature_dim
        self.feature_std = cfg.feature_std
        self.poly = cfg.poly_int_on
        # one dict per absolute level, node morton -> [8 corner ids]; corner morton -> corner id   (:47-52)
        self.node_table = [dict() for _ in range(self.max_level + 1)]
        self.corner_table = [dict() for _ in range(self.max_level + 1)]
        self.hier_features: List[torch.Tensor] = []  # top-down, [rows+1, F], trash row last  (:61-63)
        self.hierarchical_indices: List[torch.Tensor] = []  # bottom-up  (:67)
        self.importance_weight: List[torch.Tensor] = []  # (:71)
        self.features_last_frame: List[torch.Tensor] = []  # (:72)

    # model/feature_octree.py:114-166
    def update(self, surface_points: torch.Tensor, incremental_on: bool = False) -> None:
        spc = kal.unbatched_pointcloud_to_spc(surface_points, self.max_level)
        offs = spc.pyramids[0][1]
        for lvl in range(self.free_level_num, self.max_level + 1):
            nodes = spc.point_hierarchies[int(offs[lvl]) : int(offs[lvl + 1])]
            codes = kal.points_to_morton(nodes).tolist()
            known = self.node_table[lvl]
            fresh = [k for k, m in enumerate(codes) if m not in known]  # :125-127
            if not fresh:
                continue  # :129-130 (note: also skips the features_last_frame refresh)
            new_nodes = nodes[fresh]
            corners = kal.points_to_corners(new_nodes).reshape(-1, 3)  # :131
            uniq = torch.unique(corners, dim=0)  # lexicographic (x,y,z)   :132
            uniq_codes = kal.points_to_morton(uniq).tolist()
            ctab = self.corner_table[lvl]
            fl = lvl - self.free_level_num
            if len(ctab) == 0:  # first frame for this level  :135-146
                ctab.update(zip(uniq_codes, range(len(uniq_codes))))
                fts = self.feature_std * torch.randn(len(ctab) + 1, self.feature_dim)
                fts[-1] = 0.0
                self.hier_features.append(fts.requires_grad_(True))
                if incremental_on:
                    self.importance_weight.append(torch.zeros(len(ctab) + 1, self.feature_dim))
                    self.features_last_frame.append(fts.detach().clone())
            else:  # :147-160
                before = len(ctab)
                for m in uniq_codes:
                    if m not in ctab:
                        ctab[m] = len(ctab)
                added = len(ctab) - before
                tail = self.feature_std * torch.randn(added + 1, self.feature_dim)
                tail[-1] = 0.0
                grown = torch.cat((self.hier_features[fl].detach()[:-1], tail), 0)
                self.hier_features[fl] = grown.requires_grad_(True)
                if incremental_on:
                    self.importance_weight[fl] = torch.cat(
                        (self.importance_weight[fl][:-1], torch.zeros(added + 1, self.feature_dim)), 0
                    )
                    # :160 clones the *Parameter* (not .detach()'d): the copy stays attached to the graph, so in
                    # cal_regularization d/dF (F - clone(F)) = I - I = 0 — from the second frame on the
                    # regulariser adds to the loss VALUE but contributes no gradient.  Reference behaviour; kept.
                    self.features_last_frame[fl] = self.hier_features[fl].clone()
            corner_codes = kal.points_to_morton(corners).tolist()  # :162-166
            ids = [ctab[m] for m in corner_codes]
            for k, m in enumerate(kal.points_to_morton(new_nodes).tolist()):
                known[m] = ids[8 * k : 8 * k + 8]

    # model/feature_octree.py:78-81
    def set_zero(self) -> None:
        with torch.no_grad():
            for t in self.hier_features:
                t[-1] = 0.0

    # model/feature_octree.py:199-218  (the per-point dict.get loop is kept on purpose)
    def get_indices(self, coord: torch.Tensor) -> List[torch.Tensor]:
        self.hierarchical_indices = []
        for i in range(self.featured_level_num):
            lvl = self.max_level - i
            vox = kal.quantize_points(coord, lvl)
            codes = kal.points_to_morton(vox).cpu().numpy().tolist()
            tab = self.node_table[lvl]
            rows = [tab.get(m, _MISS) for m in codes]
            self.hierarchical_indices.append(torch.tensor(rows, dtype=torch.int64).reshape(-1, 8))
        return self.hierarchical_indices

    # model/feature_octree.py:172-196
    def interp_weights(self, x: torch.Tensor, level: int) -> torch.Tensor:
        u = (2 ** level) * (x * 0.5 + 0.5)
        d = torch.frac(u)
        if getattr(self, "acc_dtype", torch.float32) != torch.float32:
            d = d.to(self.acc_dtype)  # wide-accumulation mode (to_wide): fp32 coordinate arithmetic, wide sums
        if self.poly:
            tx = 3 * (d[:, 0] ** 2) - 2 * (d[:, 0] ** 3)
            ty = 3 * (d[:, 1] ** 2) - 2 * (d[:, 1] ** 3)
            tz = 3 * (d[:, 2] ** 2) - 2 * (d[:, 2] ** 3)
        else:
            tx, ty, tz = d[:, 0], d[:, 1], d[:, 2]
        ux, uy, uz = 1 - tx, 1 - ty, 1 - tz
        w = (
            ux * uy * uz,
            ux * uy * tz,
            ux * ty * uz,
            ux * ty * tz,
            tx * uy * uz,
            tx * uy * tz,
            tx * ty * uz,
            tx * ty * tz,
        )
        return torch.stack(w, 0).T.unsqueeze(2)  # [N,8,1]

    # model/feature_octree.py:222-234
    def query_feature_with_indices(self, coord: torch.Tensor, hidx: Sequence[torch.Tensor]) -> torch.Tensor:
        acc = torch.zeros(coord.shape[0], self.feature_dim, dtype=getattr(self, "acc_dtype", torch.float32))
        for i in range(self.featured_level_num):
            lvl = self.max_level - i
            fl = self.featured_level_num - i - 1
            w = self.interp_weights(coord, lvl)
            acc += (self.hier_features[fl][hidx[i]] * w).sum(1)
        return acc

    # model/feature_octree.py:237-244
    def query_feature(self, coord: torch.Tensor) -> torch.Tensor:
        self.set_zero()
        return self.query_feature_with_indices(coord, self.get_indices(coord))

    # model/feature_octree.py:246-255
    def cal_regularization(self) -> torch.Tensor:
        reg = 0.0
        for i in range(self.featured_level_num):
            fl = self.featured_level_num - i - 1
            u = self.hierarchical_indices[i].flatten().unique()
            diff = self.hier_features[fl][u] - self.features_last_frame[fl][u]
            reg = reg + (self.importance_weight[fl][u] * (diff ** 2)).sum()
        return reg

    def zero_grad(self) -> None:
        for t in self.hier_features:
            t.grad = None


# --------------------------------------------------------------------------- decoder


class OracleDecoder:
    """Restates model/decoder.py:29-63 (geo decoder, `sdf` only): Linear(F,H) ReLU Linear(H,H) ReLU Linear(H,1)."""

    NAMES = ("layers.0.weight", "layers.0.bias", "layers.1.weight", "layers.1.bias", "lout.weight", "lout.bias")

    def __init__(self, cfg, generator: Optional[torch.Generator] = None):
        F, H = cfg.feature_dim, cfg.geo_mlp_hidden_dim
        assert cfg.geo_mlp_level == 2 and cfg.geo_mlp_bias_on

        def lin(o, i):  # nn.Linear default init: U(-1/sqrt(i), 1/sqrt(i)) for both
            k = 1.0 / math.sqrt(i)
            w = (torch.rand(o, i, generator=generator) * 2 - 1) * k
            b = (torch.rand(o, generator=generator) * 2 - 1) * k
            return w.requires_grad_(True), b.requires_grad_(True)

        self.W1, self.b1 = lin(H, F)
        self.W2, self.b2 = lin(H, H)
        self.w3, self.b3 = lin(1, H)

    def params(self) -> List[torch.Tensor]:
        return [self.W1, self.b1, self.W2, self.b2, self.w3, self.b3]

    def load_state_dict(self, sd) -> None:
        with torch.no_grad():
            for p, n in zip(self.params(), self.NAMES):
                p.copy_(sd[n].to(torch.float32).cpu())

    def state_dict(self):
        return {n: p.detach().clone() for p, n in zip(self.params(), self.NAMES)}

    # model/decoder.py:49-63
    def sdf(self, feat: torch.Tensor) -> torch.Tensor:
        h = torch.relu(torch.nn.functional.linear(feat, self.W1, self.b1))
        h = torch.relu(torch.nn.functional.linear(h, self.W2, self.b2))
        return torch.nn.functional.linear(h, self.w3, self.b3).squeeze(1)

    def zero_grad(self) -> None:
        for p in self.params():
            p.grad = None


# --------------------------------------------------------------------------- losses / glue


def to_wide(octree: "OracleOctree", mlp: "OracleDecoder", dtype=torch.float64) -> None:
    """Wide-accumulation mode of the oracle, for error attribution at BASELINE sizes (tests/test_gpu_scale_parity.py):
    voxel ids and the fractional coordinates d = frac(2^l (x/2 + 1/2)) are still computed in fp32 exactly as the
    reference does (so every point lands in the same voxel with the same d), everything downstream — corner weights,
    feature sums, decoder, loss, and above all the index_put_(accumulate) / GEMM reductions of the backward pass over
    10^5..10^6 points — runs in `dtype`.  It answers "what is the exact value of the reference's function on these fp32
    inputs"; the fp32 oracle (= the reference bit for bit) and the HIP path are both compared with it."""
    octree.acc_dtype = dtype
    octree.hier_features = [t.detach().to(dtype).requires_grad_(True) for t in octree.hier_features]
    octree.importance_weight = [t.to(dtype) for t in octree.importance_weight]
    octree.features_last_frame = [t.to(dtype) for t in octree.features_last_frame]
    for n in ("W1", "b1", "W2", "b2", "w3", "b3"):
        setattr(mlp, n, getattr(mlp, n).detach().to(dtype).requires_grad_(True))


def sdf_bce_loss(pred, label, sigma, reduction="mean", weight=None):
    """utils/loss.py:17-24; `weight` is the `weighted=True` form (nn.BCEWithLogitsLoss(weight=weight): every element's
    term is scaled, the mean still divides by N).  loss_weight_on is False in all 15 shipped yamls."""
    target = torch.sigmoid(label / sigma).to(pred.dtype)
    w = None if weight is None else weight.to(pred.dtype)
    return torch.nn.functional.binary_cross_entropy_with_logits(pred, target, weight=w, reduction=reduction)


def coord_gradient(coord, pred):
    """utils/tools.py:175-185  get_gradient (create_graph=True so it is differentiable a second time)."""
    return torch.autograd.grad(
        outputs=pred, inputs=coord, grad_outputs=torch.ones_like(pred), create_graph=True, retain_graph=True
    )[0]


def sigma_sigmoid(cfg) -> float:
    """shine_batch.py:87"""
    return cfg.logistic_gaussian_ratio * cfg.sigma_sigmoid_m * cfg.scale


def train_step(octree: OracleOctree, mlp: OracleDecoder, coord, sdf_label, weight, cfg, regularize=False):
    """One inner-loop iteration minus the optimiser: shine_batch.py:115-209 / shine_incre.py:118-180.

    Returns dict(loss, pred, g, feat_grads[top-down], mlp_grads[6], indices[bottom-up]).
    """
    octree.zero_grad()
    mlp.zero_grad()
    sig = sigma_sigmoid(cfg)
    eik = bool(cfg.ekional_loss_on)
    coord = coord.detach().clone().requires_grad_(eik)
    feat = octree.query_feature(coord)
    pred = mlp.sdf(feat)
    surface = weight > 0
    g = None
    if eik:
        g = coord_gradient(coord, pred) * sig
    loss = sdf_bce_loss(pred, sdf_label, sig, cfg.loss_reduction,
                        weight=torch.abs(weight) if getattr(cfg, "loss_weight_on", False) else None)  # shine_batch.py:172-174
    parts = {"bce": loss.detach().clone()}
    if regularize:
        reg = octree.cal_regularization()
        loss = loss + cfg.lambda_forget * reg
        parts["reg"] = reg.detach().clone()
    if eik:
        e = ((1.0 - g[surface].norm(2, dim=-1)) ** 2).mean()
        loss = loss + cfg.weight_e * e
        parts["eikonal"] = e.detach().clone()
    loss.backward()
    return dict(
        loss=loss.detach(),
        parts=parts,
        pred=pred.detach(),
        feat=feat.detach(),
        g=None if g is None else g.detach(),
        feat_grads=[t.grad.clone() for t in octree.hier_features],
        mlp_grads=[p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in mlp.params()],
        indices=[t.clone() for t in octree.hierarchical_indices],
    )


def importance_sweep(octree: OracleOctree, mlp: OracleDecoder, coord_pool, label_pool, cfg, bs, down_rate=2):
    """utils/incre_learning.py:8-40  cal_feature_importance (decoder grads are produced too but ignored)."""
    sig = sigma_sigmoid(cfg)
    n = coord_pool.shape[0]
    step = bs * down_rate
    for k in range(math.ceil(n / step)):
        head, tail = k * step, min((k + 1) * step, n)
        bc = coord_pool[head:tail:down_rate]
        bl = label_pool[head:tail:down_rate]
        feat = octree.query_feature(bc)
        loss = sdf_bce_loss(mlp.sdf(feat), bl, sig, cfg.loss_reduction)
        loss.backward()
        for i in range(len(octree.importance_weight)):
            octree.importance_weight[i] += octree.hier_features[i].grad.abs()
            octree.hier_features[i].grad.zero_()
            octree.importance_weight[i][-1] *= 0


def adam_param_groups(octree: OracleOctree, mlp: OracleDecoder, lr, weight_decay=1e-7, lr_level_reduce_ratio=1.0):
    """utils/tools.py:57-83  setup_optimizer: decoder group (wd) then feature levels leaf -> coarse; Adam(0.9,0.99,eps 1e-15)."""
    groups = [{"params": mlp.params(), "lr": lr, "weight_decay": weight_decay}]
    cur = lr
    L = octree.featured_level_num
    for i in range(L):
        groups.append({"params": [octree.hier_features[L - i - 1]], "lr": cur})
        cur *= lr_level_reduce_ratio
    return torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15)


# --------------------------------------------------------------------------- sampler


def sample_along_rays(points, origin, cfg, generator=None):
    """utils/data_sampler.py:18-139 with normal/semantic labels off and behind_dropoff_on=False.

    points [M,3] (already scaled to [-1,1] space), origin [3] -> coord [M*S,3], sdf_label [M*S], weight [M*S]
    (S = surface_sample_n + free_sample_n; ray-major order; weight sign = surface(+)/free(-)).
    """
    rng = dict(generator=generator) if generator is not None else {}
    ns, nf = cfg.surface_sample_n, cfg.free_sample_n
    S = ns + nf
    rel = points - origin
    m = rel.shape[0]
    dist = torch.linalg.norm(rel, dim=1, keepdim=True)
    rng_s = cfg.surface_sample_range_m * cfg.scale
    disp_s = (torch.rand(m * ns, 1, **rng) - 0.5) * 2 * rng_s
    ratio_s = disp_s / dist.repeat(ns, 1) + 1.0
    rd = dist.repeat(nf, 1)
    hi = cfg.free_sample_end_dist_m * cfg.scale / rd + 1.0
    lo = cfg.free_sample_begin_ratio
    ratio_f = torch.rand(m * nf, 1, **rng) * (hi - lo) + lo
    disp_f = (ratio_f - 1.0) * rd
    disp = torch.cat((disp_s, disp_f), 0)
    ratio = torch.cat((ratio_s, ratio_f), 0)
    xyz = rel.repeat(S, 1) * ratio + origin
    w = torch.ones(m * S, 1)
    w[m * ns :] *= -1.0
    label = disp.squeeze(1)
    xyz = xyz.reshape(S, -1, 3).transpose(0, 1).reshape(-1, 3)
    label = label.reshape(S, -1).transpose(0, 1).reshape(-1)
    w = w.reshape(S, -1).transpose(0, 1).reshape(-1)
    return xyz.contiguous(), label.contiguous(), w.contiguous()


# ---------------------------------------------------------------------------------------------------------
# meshing query
def grid_query_coords(min_bound, max_bound, voxel_size, scale, pad_voxel=2):
    """Mesher.get_query_from_bbx (utils/mesher.py:110-152) for a box given by its two corners [m].

    Returns (coord [N,3] f32 scaled to [-1,1], voxel_num_xyz int array, voxel_origin [m]); point order x-major
    ([0,0,0],[0,0,1],...), one extra voxel layer below the box (:128-130)."""
    import numpy as np

    min_bound = np.asarray(min_bound, dtype=np.float64).copy()
    max_bound = np.asarray(max_bound, dtype=np.float64)
    num = (np.ceil((max_bound - min_bound) / voxel_size) + pad_voxel * 2).astype(np.int_)
    origin = min_bound - pad_voxel * voxel_size
    origin[2] -= voxel_size
    num[2] += 1
    x = torch.arange(int(num[0]), dtype=torch.int16)
    y = torch.arange(int(num[1]), dtype=torch.int16)
    z = torch.arange(int(num[2]), dtype=torch.int16)
    x, y, z = torch.meshgrid(x, y, z, indexing="ij")
    coord = torch.stack((x.flatten(), y.flatten(), z.flatten())).transpose(0, 1).float()
    coord *= voxel_size
    coord += torch.tensor(origin, dtype=torch.float32)
    coord *= scale
    return coord, num, origin


def mesher_query_points(octree: OracleOctree, mlp: OracleDecoder, coord, bs, mc_vis_level=1):
    """Mesher.query_points (utils/mesher.py:33-108) with query_sdf=True, query_sem=False, query_mask=True.

    sdf_pred = -Decoder.sdf(query_feature(coord, faster=True)) (:69,:92; get_indices_fast returns what get_indices
    returns), mc_mask = all(hierarchical_indices[check_level] >= 0, dim=1) (:78-86), chunked by bs: float64 numpy
    buffers when more than one chunk (:43-53), the tensors' own dtype otherwise (:90-104)."""
    import math

    import numpy as np

    n = coord.shape[0]
    iter_n = math.ceil(n / bs)
    check_level = min(octree.featured_level_num, mc_vis_level) - 1
    with torch.no_grad():
        if iter_n > 1:
            sdf_pred = np.zeros(n)
            mc_mask = np.zeros(n)
            for i in range(iter_n):
                head, tail = i * bs, min((i + 1) * bs, n)
                feat = octree.query_feature(coord[head:tail])
                sdf_pred[head:tail] = (-mlp.sdf(feat)).numpy()
                mc_mask[head:tail] = torch.all(octree.hierarchical_indices[check_level] >= 0, dim=1).numpy()
        else:
            feat = octree.query_feature(coord)
            sdf_pred = (-mlp.sdf(feat)).numpy()
            mc_mask = torch.all(octree.hierarchical_indices[check_level] >= 0, dim=1).numpy()
    return sdf_pred, mc_mask
