#!/usr/bin/env python
"""Offline experiment: what does the PHYSICAL order of the feature rows do to the fused step when the
tables do not fit the Infinity Cache?  The reference appends corner rows per frame in lexicographic (x, y, z) order
(model/feature_octree.py:132-151); the step visits nodes in Z-order.  This script re-numbers the corner rows of a built map —
   morton : rows sorted by the Morton code of the corner's coordinates (a 2x2x2 block of corners = 8 consecutive rows = 256 B)
   touch  : rows numbered in order of first use along the Z-ordered node stream (what the step's gather walks)
— rebuilds the tables with the translated ids (FeatureOctree.load_tables) and times the same launch on each layout.
    python tools/experiments/row_layout.py [kind] [frames] [azimuths] [points]"""
import copy, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.feature_octree import FeatureOctree, morton_encode, _unpack_lex
from shine_mapping_amd.sampler import SortedPool

kind = sys.argv[1] if len(sys.argv) > 1 else "kitti_large"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2800
az = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pts = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
t0 = time.time()
wl = synth.build_workload(kind, frames=frames, device="cuda", seed=42, tree_level_feat=3, azimuths=az)
torch.cuda.synchronize()
print("workload built in %.1f s: %d samples, rows %s, %.0f MB of features" % (
    time.time() - t0, wl.pool.coord.shape[0], [p.shape[0] for p in wl.octree.hier_features],
    sum(p.numel() * 4 for p in wl.octree.hier_features) / 1e6), flush=True)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
octree._sync_host()
L = octree.featured_level_num


def relabel(order_name):
    new = FeatureOctree(cfg)
    tables, feats = [], []
    ranks = octree._host_node_ranks() if order_name == "touch" else None
    for s in range(L):
        n_rows = octree._corner_count[s]
        ids = octree._node_ids[s]
        if order_name == "morton":
            lex, idl = octree._corner_lex[s], octree._corner_id_of_lex[s]
            m = morton_encode(_unpack_lex(lex))
            old_in_new_order = idl[np.argsort(m, kind="stable")]
        else:  # first use along the node stream of THIS level in the global visiting order
            stream = ids[np.argsort(ranks[s], kind="stable")].reshape(-1)
            _, first = np.unique(stream, return_index=True)
            old_in_new_order = stream[np.sort(first)]
        assert old_in_new_order.size == n_rows
        new_of_old = np.empty(n_rows, np.int64)
        new_of_old[old_in_new_order] = np.arange(n_rows)
        tables.append((torch.from_numpy(octree._node_keys[s]), torch.from_numpy(new_of_old[ids].astype(np.int32))))
        f = octree.hier_features[s].detach()
        nf = torch.empty_like(f)
        nf[:-1] = f[:-1][torch.from_numpy(old_in_new_order).to(f.device)]
        nf[-1] = 0
        feats.append(torch.nn.Parameter(nf))
    new.load_tables(tables)
    new.hier_features = torch.nn.ParameterList(feats)
    return new


def time_step(oc, tag, variants=(0,)):
    params = list(oc.hier_features) + dec.fused_params()
    for p in params:
        p.grad = torch.zeros_like(p)
    oc._require_tables(with_ranks=True)
    sp = SortedPool(oc, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
    idx = sp.draw(pts)
    ns = (sp.weight[idx.long()] > 0).sum() if cfg.ekional_loss_on else None
    out = {}
    for v in variants:
        o = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, kernel_variant=0x2000 | v)
        ts = []
        for rep in range(3):
            for _ in range(3):
                fused_train_step(oc, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fused_train_step(oc, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        out[v] = ts
    o1 = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e)
    for p in params:
        p.grad.zero_()
    loss, pred, _ = fused_train_step(oc, dec, None, None, None, o1, n_surf=ns, pool=sp, idx=idx)
    print("%-8s loss %.9g  kernel us by variant %s" % (tag, float(loss), {k: ["%.1f" % t for t in v] for k, v in out.items()}), flush=True)
    del sp
    for p in params:
        p.grad = None
    torch.cuda.empty_cache()


VAR = tuple(int(v) for v in os.environ.get("LAYOUT_VARIANTS", "6,5").split(","))
time_step(octree, "appended", VAR)
for name in ("morton", "touch"):
    t0 = time.time()
    oc = relabel(name)
    print("relabel %s: %.1f s" % (name, time.time() - t0), flush=True)
    time_step(oc, name, VAR)
    del oc
