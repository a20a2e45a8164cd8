#!/usr/bin/env python
"""A/B timing of fused-step debug variants (StepOptions.kernel_variant bits, see shine_step_v1.hip) inside ONE process on
ONE box — box-to-box variation on the pool is larger than most single-change effects.

    python tools/ab_variant.py 0x2000 0x2100      # e.g. fused kernel alone vs. the same without atomics
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [int(v, 0) for v in sys.argv[1:]] or [0x2000, 0x2000]
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.sampler import SortedPool
for kind, pts, lv in (("maicity", 1 << 18, 4), ("kitti", 1 << 20, 3)):
    wl = synth.build_workload(kind, frames=60, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    params = list(octree.hier_features) + dec.fused_params()
    for p in params: p.grad = torch.zeros_like(p)
    octree._require_tables(with_ranks=True)
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
    idx = sp.draw(pts)
    ns = (sp.weight[idx.long()] > 0).sum() if cfg.ekional_loss_on else None
    res = {}
    for rep in range(3):
        for name, var in [("0x%x#%d" % (v, i), v) for i, v in enumerate(VARIANTS)]:
            o = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, kernel_variant=var)
            for _ in range(5): fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(kind, {k: ["%.1f" % v for v in vs] for k, vs in res.items()})
