#!/usr/bin/env python
"""The fused kernel of the LOADED library alone on a bench workload.  With measurement builds (tools/mk_variant.py NAME
-DSHINE_V3_ABLATE=bits: 1 no feature-grad atomics, 8 no row gathers; tools/run_with_lib.py) it says how much of the kernel's time
a part holds EXCLUSIVELY, i.e. how well the parts overlap.   python tools/far_ablate.py [kitti-large|kitti|maicity]"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, benchlib, fused_train_step, synth
from shine_mapping_amd.sampler import SortedPool

name = sys.argv[1] if len(sys.argv) > 1 else "kitti-large"
spec = benchlib.WORKLOADS[name]
wl = synth.build_workload(spec["preset"], frames=spec["frames"], device="cuda", seed=42, tree_level_feat=spec["levels"], azimuths=spec["azimuths"])
cfg, octree, dec = wl.cfg, wl.octree, wl.decoder
octree._require_tables(with_ranks=True)
for p in list(octree.hier_features) + dec.fused_params():
    p.grad = torch.zeros_like(p)
sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=1000)
n = spec["points"]
idx = sp.draw(n)
eik = bool(cfg.ekional_loss_on)
ns = (sp.weight[idx.long()] > 0).sum() if eik else None
base = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction, ekional_loss_on=eik, weight_e=cfg.weight_e)
print("%s: %d points, L=%d, eikonal %s" % (name, n, cfg.tree_level_feat, eik))
for label, bits in (("kernel of the loaded library", 0),):
    o = copy.copy(base)
    o.kernel_variant = 0x2000 | (bits << 8)
    for _ in range(5):
        fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print("  %-26s %7.1f us" % (label, sorted(ts)[2]), flush=True)
