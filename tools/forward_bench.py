#!/usr/bin/env python
"""Forward-only throughput (Mesher.query_points style: query_feature + sdf on grid-ordered points, utils/mesher.py:33-108)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import forward_sdf, synth

wl = synth.build_workload("maicity", frames=20, device="cuda", seed=42, tree_level_feat=3)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
# a regular grid over part of the street at 0.1 m (mc_res_m), x fastest -> spatially coherent like the mesher's
n_side = (400, 160, 60)
gx = torch.arange(n_side[0], device="cuda") * 0.1 - 20.0
gy = torch.arange(n_side[1], device="cuda") * 0.1 - 8.0
gz = torch.arange(n_side[2], device="cuda") * 0.1 - 0.5
grid = torch.stack(torch.meshgrid(gz, gy, gx, indexing="ij"), -1).reshape(-1, 3)[:, [2, 1, 0]].contiguous() * cfg.scale
n = grid.shape[0]
for want_idx in (False, True):
    for _ in range(3):
        out = forward_sdf(octree, dec, grid, want_indices=want_idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = forward_sdf(octree, dec, grid, want_indices=want_idx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("forward_sdf N=%d indices=%s: %.3f ms  %.2f G pts/s" % (n, want_idx, dt * 1e3, n / dt / 1e9))
