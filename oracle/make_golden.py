"""Generate tests/golden/*.pt by running the REAL reference (from /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the authoring container:

    python -m oracle.make_golden            # writes tests/golden/*.pt, checks the oracle bit-for-bit

The reference's own FeatureOctree / Decoder / sdf_bce_loss / get_gradient /
dataSampler / cal_feature_importance are executed unmodified (kaolin's five
integer ops come from oracle/kaolin_shim.py; see oracle/ref_import.py for the
inert import stubs).  For every case the same inputs are pushed through
oracle/shine_oracle.py and the outputs must be bit-identical, which pins the
restatement to the reference on this machine; the saved fixtures pin it (and
the HIP path) everywhere else.  /root/reference cannot travel to the GPU box,
the fixtures can.
"""
import os
import sys

import torch

from . import ref_import
from . import shine_oracle as so

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (config overrides, N, decoder init, feature gain, regularize, second frame)
    "maicity_bce_L3": dict(cfg=dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.2, sigma_sigmoid_m=0.05,
                                    surface_sample_range_m=0.15, surface_sample_n=3, free_sample_n=3,
                                    free_sample_end_dist_m=0.8), n=2048, decoder="pretrained", gain=1.0),
    "maicity_bce_L4": dict(cfg=dict(tree_level_world=12, tree_level_feat=4, leaf_vox_size=0.2, sigma_sigmoid_m=0.05,
                                    surface_sample_range_m=0.15, surface_sample_n=3, free_sample_n=3,
                                    free_sample_end_dist_m=0.8), n=2048, decoder="random", gain=4.0),
    "kitti_eik_L3": dict(cfg=dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.3, sigma_sigmoid_m=0.1,
                                  surface_sample_range_m=0.3, surface_sample_n=3, free_sample_n=3,
                                  free_sample_end_dist_m=0.8, ekional_loss_on=True, weight_e=0.1),
                         n=2048, decoder="pretrained", gain=10.0),
    "ncd_reg_L3": dict(cfg=dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.2, sigma_sigmoid_m=0.1,
                                surface_sample_range_m=0.3, surface_sample_n=3, free_sample_n=3,
                                free_sample_end_dist_m=1.0, loss_reduction="sum", lambda_forget=1e4),
                       n=1024, decoder="pretrained", gain=2.0, incremental=True),
    "linear_L2_nopoly": dict(cfg=dict(tree_level_world=8, tree_level_feat=2, leaf_vox_size=0.4, sigma_sigmoid_m=0.1,
                                      poly_int_on=False, ekional_loss_on=True, weight_e=0.05,
                                      surface_sample_range_m=0.3, surface_sample_n=2, free_sample_n=2,
                                      free_sample_end_dist_m=0.8), n=1024, decoder="random", gain=8.0),
}


def tiny_scan(seed, n_pts, shift=0.0):
    """A ground patch and a wall seen from one pose, metres. Deterministic."""
    g = torch.Generator().manual_seed(seed)
    half = n_pts // 2
    ground = torch.stack((torch.rand(half, generator=g) * 12 - 6 + shift, torch.rand(half, generator=g) * 8 - 4,
                          torch.zeros(half)), 1)
    wall = torch.stack((torch.rand(n_pts - half, generator=g) * 12 - 6 + shift, torch.full((n_pts - half,), 4.0),
                        torch.rand(n_pts - half, generator=g) * 3), 1)
    origin = torch.tensor([shift, 0.0, 1.8])
    return torch.cat((ground, wall), 0), origin


def ref_config(R, over):
    c = R.SHINEConfig()
    c.device = "cpu"
    for k, v in over.items():
        setattr(c, k, v)
    c.scale = 1.0 / (c.leaf_vox_size * (2 ** (c.tree_level_world - 1)))  # utils/config.py:372-374
    return c


def tables_of(octree):
    """Per featured level (top-down): node mortons [n] int64 and corner ids [n,8] int32, insertion order."""
    out = []
    for lvl in range(octree.free_level_num, octree.max_level + 1):
        tab = octree.nodes_lookup_tables[lvl]
        keys = torch.tensor(list(tab.keys()), dtype=torch.int64)
        vals = torch.tensor(list(tab.values()), dtype=torch.int32).reshape(-1, 8)
        out.append((keys, vals))
    return out


def run_case(R, name, spec):
    torch.manual_seed(1234)
    cfg = ref_config(R, spec["cfg"])
    ocfg = so.make_config(**spec["cfg"])
    assert ocfg.scale == cfg.scale
    sampler = R.dataSampler(cfg)
    octree = R.FeatureOctree(cfg)
    mlp = R.Decoder(cfg)
    if spec["decoder"] == "pretrained":
        sd = torch.load(os.path.join(ref_import.REFERENCE_ROOT, "pretrained", "geo_decoder_8dim.pth"),
                        map_location="cpu")["geo_decoder"]
        mlp.load_state_dict(sd)
    incremental = spec.get("incremental", False)

    oct2 = so.OracleOctree(ocfg)
    mlp2 = so.OracleDecoder(ocfg)
    mlp2.load_state_dict(mlp.state_dict())

    frames = []
    for f in range(2 if incremental else 1):
        pts, origin = tiny_scan(100 + f, 1500, shift=2.5 * f)
        rs = torch.random.get_rng_state()
        coord, label, _, _, weight, _, _ = sampler.sample(pts * cfg.scale, origin * cfg.scale, None, None)
        torch.random.set_rng_state(rs)
        c2, l2, w2 = so.sample_along_rays(pts * cfg.scale, origin * cfg.scale, ocfg)
        assert torch.equal(coord, c2) and torch.equal(label, l2) and torch.equal(weight, w2), "sampler restatement drifted"
        rs = torch.random.get_rng_state()
        octree.update(coord[weight > 0, :], incremental)
        torch.random.set_rng_state(rs)
        oct2.update(coord[weight > 0, :], incremental)
        frames.append((coord, label, weight))
        if incremental and f == 0:
            # pretend a first frame was trained: move features, accumulate a non-trivial importance
            with torch.no_grad():
                for i, p in enumerate(octree.hier_features):
                    p[:-1] += 0.03 * torch.randn_like(p[:-1])
                    oct2.hier_features[i].copy_(p)
            pool = SimpleDataset(coord, label)
            R.cal_feature_importance(pool, octree, mlp, cfg.logistic_gaussian_ratio * cfg.sigma_sigmoid_m * cfg.scale,
                                     256, 2, cfg.loss_reduction)
            so.importance_sweep(oct2, mlp2, coord, label, ocfg, 256, 2)
            for a, b in zip(octree.importance_weight, oct2.importance_weight):
                assert torch.equal(a, b), "importance sweep restatement drifted"
            for p in octree.hier_features:
                p.grad = None
            for p in mlp.parameters():
                p.grad = None
    # identical tables / rows?
    for lvl in range(octree.max_level + 1):
        assert octree.nodes_lookup_tables[lvl] == oct2.node_table[lvl], "octree build restatement drifted"
    with torch.no_grad():
        for i, p in enumerate(octree.hier_features):
            assert torch.equal(p, oct2.hier_features[i])
            p[:-1] *= spec["gain"]
            oct2.hier_features[i][:-1] *= spec["gain"]
        if incremental:  # drift from last frame so the regulariser is non-zero
            for i, p in enumerate(octree.hier_features):
                d = 0.01 * torch.randn_like(p[:-1])
                p[:-1] += d
                oct2.hier_features[i][:-1] += d

    coord, label, weight = frames[-1]
    n = spec["n"]
    idx = torch.randint(0, coord.shape[0], (n,))
    bc, bl, bw = coord[idx].clone(), label[idx].clone(), weight[idx].clone()
    # a few deliberately awkward points: exactly on voxel faces, outside the cube, far from any node
    bc[0] = torch.tensor([0.0, 0.0, 0.0])
    bc[1] = torch.tensor([1.0, -1.0, 0.5])
    bc[2] = torch.tensor([1.5, -1.25, 0.0])
    bc[3] = bc[4]

    # ---- the reference's inner loop, shine_batch.py:115-209 / shine_incre.py:118-180
    sig = cfg.logistic_gaussian_ratio * cfg.sigma_sigmoid_m * cfg.scale
    eik = cfg.ekional_loss_on
    rc = bc.clone().requires_grad_(eik)
    feat = octree.query_feature(rc)
    pred = mlp.sdf(feat)
    surface = bw > 0
    g = R.get_gradient(rc, pred) * sig if eik else None
    w_abs = torch.abs(bw)
    loss = R.sdf_bce_loss(pred, bl, sig, w_abs, cfg.loss_weight_on, cfg.loss_reduction)
    parts = {"bce": loss.detach().clone()}
    if incremental:
        reg = octree.cal_regularization()
        loss = loss + cfg.lambda_forget * reg
        parts["reg"] = reg.detach().clone()
    if eik:
        e = ((1.0 - g[surface].norm(2, dim=-1)) ** 2).mean()
        loss = loss + cfg.weight_e * e
        parts["eikonal"] = e.detach().clone()
    loss.backward()
    ref_out = dict(
        loss=loss.detach().clone(), parts=parts, pred=pred.detach().clone(), feat=feat.detach().clone(),
        g=None if g is None else g.detach().clone(),
        feat_grads=[p.grad.clone() for p in octree.hier_features],
        mlp_grads=[mlp.layers[0].weight.grad.clone(), mlp.layers[0].bias.grad.clone(), mlp.layers[1].weight.grad.clone(),
                   mlp.layers[1].bias.grad.clone(), mlp.lout.weight.grad.clone(), mlp.lout.bias.grad.clone()],
        indices=[t.clone() for t in octree.hierarchical_indices],
    )

    # ---- the oracle on the same inputs must agree to the bit
    out = so.train_step(oct2, mlp2, bc, bl, bw, ocfg, regularize=incremental)
    def same(a, b, what):
        assert torch.equal(a, b), "%s: oracle != reference in %s (max |d| %g)" % (name, what, (a - b).abs().max())
    same(out["pred"], ref_out["pred"], "pred")
    same(out["feat"], ref_out["feat"], "feat")
    same(out["loss"], ref_out["loss"], "loss")
    if eik:
        same(out["g"], ref_out["g"], "g")
    for k in range(len(ref_out["feat_grads"])):
        same(out["feat_grads"][k], ref_out["feat_grads"][k], "feat_grad[%d]" % k)
    for k in range(6):
        same(out["mlp_grads"][k], ref_out["mlp_grads"][k], "mlp_grad[%d]" % k)
    for k in range(len(ref_out["indices"])):
        same(out["indices"][k], ref_out["indices"][k], "indices[%d]" % k)

    fixture = dict(
        name=name, cfg=dict(spec["cfg"]), scale=cfg.scale, sigma=sig, regularize=incremental,
        tables=tables_of(octree),
        features=[p.detach().clone() for p in octree.hier_features],
        importance=[t.clone() for t in octree.importance_weight] if incremental else None,
        features_last=[t.clone() for t in octree.features_last_frame] if incremental else None,
        decoder={k: v.detach().clone() for k, v in mlp.state_dict().items() if not k.startswith("nclass_out")},
        coord=bc, sdf_label=bl, weight=bw,
        surface_points=[fr[0][fr[2] > 0].clone() for fr in frames],
        out=ref_out,
        provenance="reference @ /root/reference executed on CPU by oracle/make_golden.py, torch %s" % torch.__version__,
    )
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fixture, path)
    print("%-18s N=%d rows=%s loss=%.6f -> %s (%.0f kB)" % (
        name, n, [int(p.shape[0]) for p in octree.hier_features], float(loss), os.path.relpath(path),
        os.path.getsize(path) / 1024))


MESH_CASES = {
    # name: source fixture, box [m], marching-cubes voxel [m], chunk size, mc_vis_level
    "mesh_query_L3": dict(source="maicity_bce_L3", lo=(-6.0, -4.0, 0.0), hi=(6.0, 4.3, 3.0), voxel=0.17, bs=20000, vis=1),
    "mesh_query_L4": dict(source="maicity_bce_L4", lo=(-5.0, -3.0, 0.0), hi=(5.0, 4.2, 2.5), voxel=0.23, bs=10 ** 9, vis=2),
}


class _Box:
    """Duck-typed open3d AxisAlignedBoundingBox: get_query_from_bbx only reads the two corners (utils/mesher.py:124-125)."""

    def __init__(self, lo, hi):
        import numpy as np
        self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)

    def get_min_bound(self):
        return self.lo.copy()

    def get_max_bound(self):
        return self.hi.copy()


def run_mesh_case(R, name, spec):
    """The reference's Mesher.get_query_from_bbx + query_points on a fixture's map (utils/mesher.py:33-152)."""
    import numpy as np

    fx = torch.load(os.path.join(GOLDEN_DIR, spec["source"] + ".pt"), weights_only=False)
    cfg = ref_config(R, fx["cfg"])
    cfg.mc_vis_level = spec["vis"]
    octree = R.FeatureOctree(cfg)
    for s, (keys, ids) in enumerate(fx["tables"]):
        octree.nodes_lookup_tables[octree.free_level_num + s] = dict(zip(keys.tolist(), ids.tolist()))
    octree.hier_features = torch.nn.ParameterList([torch.nn.Parameter(f.clone()) for f in fx["features"]])
    mlp = R.Decoder(cfg)
    mlp.load_state_dict(fx["decoder"], strict=False)
    mesher = R.Mesher(cfg, octree, mlp, None)
    coord, num, origin = mesher.get_query_from_bbx(_Box(spec["lo"], spec["hi"]), spec["voxel"])
    sdf_pred, sem_pred, mc_mask = mesher.query_points(coord, spec["bs"], True, False, True)
    assert sem_pred is None

    ocfg = so.make_config(**fx["cfg"])
    oct2 = so.OracleOctree(ocfg)
    for s, (keys, ids) in enumerate(fx["tables"]):
        oct2.node_table[oct2.free_level_num + s] = dict(zip(keys.tolist(), ids.tolist()))
    oct2.hier_features = [f.clone() for f in fx["features"]]
    mlp2 = so.OracleDecoder(ocfg)
    mlp2.load_state_dict(fx["decoder"])
    c2, n2, o2 = so.grid_query_coords(spec["lo"], spec["hi"], spec["voxel"], ocfg.scale, cfg.pad_voxel)
    assert torch.equal(coord, c2) and np.array_equal(num, n2) and np.array_equal(origin, o2), "grid restatement drifted"
    s2, m2 = so.mesher_query_points(oct2, mlp2, c2, spec["bs"], spec["vis"])
    assert sdf_pred.dtype == s2.dtype and np.array_equal(sdf_pred, s2), "mesher query restatement drifted (sdf)"
    assert mc_mask.dtype == m2.dtype and np.array_equal(mc_mask, m2), "mesher query restatement drifted (mask)"

    fixture = dict(name=name, source=spec["source"], lo=spec["lo"], hi=spec["hi"], voxel=spec["voxel"], bs=spec["bs"],
                   mc_vis_level=spec["vis"], pad_voxel=cfg.pad_voxel, voxel_num_xyz=torch.from_numpy(num.astype("int64")),
                   voxel_origin=torch.from_numpy(origin), n=int(coord.shape[0]),
                   sdf_dtype=str(sdf_pred.dtype), mask_dtype=str(mc_mask.dtype),
                   sdf_pred=torch.from_numpy(sdf_pred.astype("float32")),  # values are fp32 either way: lossless
                   mc_mask=torch.from_numpy(mc_mask.astype("uint8")),
                   provenance="reference Mesher @ /root/reference on CPU by oracle/make_golden.py, torch %s" % torch.__version__)
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    torch.save(fixture, path)
    print("%-18s N=%d grid=%s masked-in=%d -> %s (%.0f kB)" % (name, coord.shape[0], num.tolist(), int(mc_mask.sum()),
                                                              os.path.relpath(path), os.path.getsize(path) / 1024))


class SimpleDataset:
    """Duck-typed stand-in for LiDARDataset: cal_feature_importance only reads the two pools (utils/incre_learning.py:14-26)."""

    def __init__(self, coord, label):
        self.coord_pool = coord
        self.sdf_label_pool = label


def main():
    if not ref_import.available():
        sys.exit("needs /root/reference (authoring container only)")
    R = ref_import.install()
    torch.set_num_threads(1)  # bit-stable reductions
    only_mesh = "--mesh-only" in sys.argv  # leaves the step fixtures untouched
    if not only_mesh:
        for name, spec in CASES.items():
            run_case(R, name, spec)
    for name, spec in MESH_CASES.items():
        run_mesh_case(R, name, spec)


if __name__ == "__main__":
    main()
