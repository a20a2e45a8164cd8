#!/usr/bin/env python
"""An UNORDERED 2^18-point batch (what the reference's get_batch returns) through fused_train_step: in-kernel probing
(auto_plan_min=0) vs plan + planned-batch kernel (the default)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth

for kind, pts, lv in (("maicity", 1 << 18, 4), ("kitti", 1 << 20, 3), ("maicity", 4096, 4), ("maicity", 16384, 4),
                      ("maicity", 32768, 4), ("maicity", 65536, 4), ("maicity", 131072, 4), ("kitti", 65536, 3)):
    wl = synth.build_workload(kind, frames=60, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    for p in list(octree.hier_features) + dec.fused_params():
        p.grad = torch.zeros_like(p)
    g = torch.Generator(device="cuda").manual_seed(1)
    c, l, w = synth.draw_batch(wl.pool, pts, g)
    out = {}
    for name, apm in (("probing", 0), ("auto-plan", 1)):
        o = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, auto_plan_min=apm)
        for _ in range(5):
            fused_train_step(octree, dec, c, l, w, o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fused_train_step(octree, dec, c, l, w, o)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 20 * 1e3
    print(kind, pts, lv, {k: "%.1f us" % v for k, v in out.items()})
