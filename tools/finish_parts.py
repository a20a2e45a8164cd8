#!/usr/bin/env python
"""The iteration tail (shine_finish_iteration, k_finish) at N = 4096 taken apart: the full launch against the launch without the
next draw, without the regulariser, with a dense (not active-row) Adam — each as a torch-captured graph of 50 launches on the
partial sums of one fused step.     python tools/finish_parts.py [ncd|maicity] [levels]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.loop import GraphedIteration
from shine_mapping_amd.optim import setup_optimizer
from shine_mapping_amd.sampler import SortedPool

kind = sys.argv[1] if len(sys.argv) > 1 else "ncd"
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = synth.build_workload(kind, frames=12, device="cuda", seed=42, tree_level_feat=lv, azimuths=450)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
octree._require_tables(with_ranks=True)
n = 4096
lam = float(getattr(cfg, "lambda_forget", 0.0)) or 1e4
if not octree.importance_weight:
    octree.importance_weight = [torch.rand_like(p.detach()) for p in octree.hier_features]
    octree.features_last_frame = [p.detach().clone() for p in octree.hier_features]
    octree._reg_grad_on = [True] * len(octree.hier_features)
res = {}
for name, kw in (("full (active rows, regulariser, next draw)", {}), ("no next draw", {"draw": False}), ("no regulariser", {"reg": False}),
                 ("dense Adam (no active rows)", {"active": False}), ("Adam only (dense, no regulariser, no draw)", {"draw": False, "reg": False, "active": False})):
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=3)
    opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum")
    it = GraphedIteration(octree, dec, sp, opt, opts, n, lambda_forget=lam if kw.get("reg", True) else 0.0, native=False,
                          active_rows=kw.get("active", True))
    it.run(8)  # some rows touched, optimiser state present
    torch.cuda.synchronize()
    pending = {}
    fused_train_step(octree, dec, None, None, None, it._hooked, pool=sp, idx=it._idx, touched=it.touched, pending=pending)

    def tail():
        opt.finish_iteration(pending, dict(lambda_forget=lam, touched=it.touched, out=it._reg_out) if it.regularize else None,
                             next_draw=sp.next_draw(n, it._idx, None) if kw.get("draw", True) else None,
                             active_flags=it.touched if it.active_rows else None)

    for _ in range(3):
        tail()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            tail()
    g.replay()
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    res[name] = ["%.2f" % t for t in ts]
    print(kind, "L%d" % lv, [int(p.shape[0]) for p in octree.hier_features], name, res[name], "us per launch", flush=True)
