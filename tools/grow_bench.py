#!/usr/bin/env python
"""Per-frame cost of FeatureOctree.update (model/feature_octree.py:114-166): device growth (shine_tables_grow) vs
the vectorised host path, on the synthetic MaiCity-like frames bench.py uses.  (The CPU-oracle baseline for the same
frames is timed by tests/cpu_baselines.py: only tests/ may touch oracle/.)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import FeatureOctree, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = synth.make_config("maicity", device="cuda", tree_level_feat=3)
data = [(c[w > 0].contiguous()) for c, l, w in synth.make_frames(cfg, frames, 64, 1800, 42, "cuda")]
print("frames=%d surface points/frame ~%d" % (frames, int(np.mean([d.shape[0] for d in data]))))


def run(move):
    torch.manual_seed(0)
    octree = FeatureOctree(cfg)
    times = []
    for d in data:
        p = move(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        octree.update(p, True)
        octree._require_tables(with_ranks=True)  # what the next training iteration needs: tables + node ranks
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return octree, np.array(times) * 1e3


for name, move in (("device (shine_tables_grow)", lambda d: d), ("host (numpy)", lambda d: d.cpu())):
    run(move)  # warm-up (allocator, rocPRIM kernels)
    octree, t = run(move)
    print("%-28s first frame %.2f ms, median later frames %.2f ms, total %.1f ms; rows %s" % (
        name, t[0], float(np.median(t[1:])), t.sum(), [int(p.shape[0]) for p in octree.hier_features]))
