#!/usr/bin/env python
"""A/B of the fused-step kernels inside one process: HIP-event time of the dominant kernel alone (reduction launch
skipped) per kernel_variant (2: 32-point tiles, 3: 16-point tiles), per workload / batch size.

    python tools/ab_kernel.py [--workload maicity] [--levels 4] [--points 262144,4096] [--variants 2,3]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.sampler import SortedPool

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="maicity")
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--points", default="262144,4096")
ap.add_argument("--variants", default="2,3")
ap.add_argument("--frames", type=int, default=60)
args = ap.parse_args()
wl = synth.build_workload(args.workload, frames=args.frames, device="cuda", seed=42, tree_level_feat=args.levels)
cfg, octree, dec = wl.cfg, wl.octree, wl.decoder
octree._require_tables(with_ranks=True)
sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=1)
for p in list(octree.hier_features) + dec.fused_params():
    p.grad = torch.zeros_like(p)
for n in [int(x) for x in args.points.split(",")]:
    idx = sp.draw(n)
    ns = (sp.weight[idx.long()] > 0).sum() if cfg.ekional_loss_on else None
    for v in [int(x) for x in args.variants.split(",")]:
        opts = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e,
                           kernel_variant=0x2000 | v)
        try:
            for _ in range(5):
                fused_train_step(octree, dec, None, None, None, opts, n_surf=ns, pool=sp, idx=idx)
        except Exception as e:
            print("n=%d variant %d: %s" % (n, v, e))
            continue
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()  # replayed from a HIP graph: the host's launch cost must not bound the measurement
        with torch.cuda.graph(g):
            for _ in range(10):
                fused_train_step(octree, dec, None, None, None, opts, n_surf=ns, pool=sp, idx=idx)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        print("%s L%d n=%d variant %d: kernel %.1f us (min %.1f)" % (args.workload, args.levels, n, v, sorted(ts)[3], min(ts)))
