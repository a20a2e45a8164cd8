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

# the meshing kernel proper (csrc/shine_query.hip): reference grid order (z fastest, utils/mesher.py:139-141) and
# the reference's per-chunk contract (sdf + marching-cubes mask), timed with HIP events on the current stream
from shine_mapping_amd.mesher import Mesher


class _Box:
    def get_min_bound(self):
        import numpy as np
        return np.array([-20.0, -8.0, -0.5])

    def get_max_bound(self):
        import numpy as np
        return np.array([20.0, 8.0, 5.5])


mesher = Mesher(cfg, octree, dec, None)
coord, num, origin = mesher.get_query_from_bbx(_Box(), 0.1)
from shine_mapping_amd.mesher import query_points_device
def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("  mask only (no decoder): %.3f ms;  sdf only: %.3f ms;  random order sdf+mask: %.3f ms" % (
    timed(lambda: query_points_device(octree, dec, coord, query_sdf=False)),
    timed(lambda: query_points_device(octree, dec, coord, query_mask=False)),
    timed(lambda c=coord[torch.randperm(coord.shape[0], device="cuda")]: query_points_device(octree, dec, c))))
for _ in range(3):
    sdf, mask = query_points_device(octree, dec, coord)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    sdf, mask = query_points_device(octree, dec, coord)
e1.record()
torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / 10 * 1e-3
print("shine_query_points N=%d grid=%s: %.3f ms  %.2f G pts/s  (mask-in %.1f %%)" % (
    coord.shape[0], num.tolist(), dt * 1e3, coord.shape[0] / dt / 1e9, 100.0 * mask.float().mean().item()))
