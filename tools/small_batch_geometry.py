#!/usr/bin/env python
"""the fused kernel at the reference's batch size (N = 4096 = 256 tiles) as 64 workgroups x 4 waves (the
product: 64 CUs) against 256 workgroups with ONE active wave each (kernel_variant bit 0x8000: every CU) — kernel alone,
back-to-back launches inside one HIP graph (no launch gaps), on the maicity-like and the ncd-like map.
    python tools/small_batch_geometry.py"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.sampler import SortedPool

for kind, levels, frames, az in (("maicity", 3, 30, 450), ("ncd", 3, 24, 900)):
    wl = synth.build_workload(kind, frames=frames, device="cuda", seed=42, tree_level_feat=levels, azimuths=az)
    cfg, octree, dec = wl.cfg, wl.octree, wl.decoder
    octree._require_tables(with_ranks=True)
    for p in list(octree.hier_features) + dec.fused_params():
        p.grad = torch.zeros_like(p)
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=5)
    for n in (4096, 8192, 16384):
        idx = sp.draw(n)
        base = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction)
        out = []
        for label, bits in (("four tiles per workgroup (product)", 0x2000), ("one tile per workgroup", 0xA000)):
            o = copy.copy(base)
            o.kernel_variant = bits
            R = 20
            for _ in range(3):
                fused_train_step(octree, dec, None, None, None, o, pool=sp, idx=idx)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(R):
                    fused_train_step(octree, dec, None, None, None, o, pool=sp, idx=idx)
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / R * 1e3)
            out.append("%s %.2f us" % (label, sorted(ts)[3]))
        print("%s L%d N=%d: %s" % (kind, levels, n, "; ".join(out)), flush=True)
