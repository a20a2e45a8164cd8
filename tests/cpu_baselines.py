#!/usr/bin/env python
"""CPU-oracle timings quoted in DESIGN.md next to the GPU numbers of tools/grow_bench.py and tools/incre_bench.py
(test infrastructure: lives under tests/ because only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/).

    python tests/cpu_baselines.py grow        # OracleOctree.update per frame (the reference's Python loops restated)
    python tests/cpu_baselines.py incre       # one incremental-mapping frame: update, iterations, importance sweep

Needs no GPU (frames are generated on the CPU); run it on the GPU box to time the same host cores bench.py reports.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shine_oracle as so  # noqa: E402
from shine_mapping_amd import synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "grow"
torch.set_num_threads(1)
if what == "grow":
    cfg = synth.make_config("maicity", device="cpu", tree_level_feat=3)
    ocfg = so.make_config(tree_level_world=cfg.tree_level_world, tree_level_feat=3, leaf_vox_size=cfg.leaf_vox_size)
    ref = so.OracleOctree(ocfg)
    for i, (c, l, w) in enumerate(synth.make_frames(cfg, 3, 64, 1800, 42, "cpu")):
        pts = c[w > 0]
        t0 = time.perf_counter()
        ref.update(pts, True)
        print("oracle update, frame %d (%d surface points): %.0f ms" % (i, pts.shape[0], (time.perf_counter() - t0) * 1e3))
else:
    cfg = synth.make_config("ncd", device="cpu")
    over = dict(tree_level_world=cfg.tree_level_world, tree_level_feat=cfg.tree_level_feat, leaf_vox_size=cfg.leaf_vox_size,
                sigma_sigmoid_m=cfg.sigma_sigmoid_m, loss_reduction="sum", lambda_forget=cfg.lambda_forget)
    ocfg = so.make_config(**over)
    ref, mlp = so.OracleOctree(ocfg), so.OracleDecoder(ocfg)
    bs, iters = 4096, 50
    for fi, (c, l, w) in enumerate(synth.make_frames(cfg, 2, 64, 900, 42, "cpu")):
        t0 = time.perf_counter()
        ref.update(c[w > 0], True)
        t1 = time.perf_counter()
        opt = so.adam_param_groups(ref, mlp, 0.01)
        n_it = 10
        for it in range(n_it):
            idx = torch.randint(0, c.shape[0], (bs,))
            so.train_step(ref, mlp, c[idx], l[idx], w[idx], ocfg, regularize=True)
            opt.step()
            opt.zero_grad(set_to_none=True)
        t2 = time.perf_counter()
        so.importance_sweep(ref, mlp, c, l, ocfg, bs, 2)
        t3 = time.perf_counter()
        print("oracle frame %d: update %.0f ms | iteration %.1f ms (x%d = %.0f ms) | importance sweep %.0f ms" % (
            fi, (t1 - t0) * 1e3, (t2 - t1) / n_it * 1e3, iters, (t2 - t1) / n_it * iters * 1e3, (t3 - t2) * 1e3))
