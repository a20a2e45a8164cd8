#!/usr/bin/env python
"""How many feature-grad atomics does the headline step need?  For one sorted draw of the bench's workload, per level: node runs
of the ordered stream (the product's count: one 64-lane atomic = 8 row-atomics per run) against the number of DISTINCT corner
rows per window of W consecutive points (what merging by corner inside a tile / a wave's range / a workgroup's range would issue).

    python tools/atomics_count.py [maicity|kitti|kitti_large] [points] [levels] [frames] [azimuths]
(kitti_large 1048576 3 2800 300 = bench.py's kitti-large: windows of 512 points = one wave's tile range of the full-chip launch,
4096 = one workgroup's)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import synth  # noqa: E402
from shine_mapping_amd.sampler import SortedPool  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "maicity"
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
lv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 60
az = int(sys.argv[5]) if len(sys.argv) > 5 else 450
wl = synth.build_workload(kind, frames=frames, device="cuda", seed=42, tree_level_feat=lv, azimuths=az)
octree = wl.octree
octree._require_tables(with_ranks=True)
sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
idx = sp.draw(pts).long()
coord = sp.coord[idx]
hi = octree.get_indices(coord)  # bottom-up list of [N, 8] corner ids (-1: miss)
tot_runs = tot_rows = 0
print("%s, %d points, %d levels; row-atomics (32 B each) per step" % (kind, pts, lv))
for k, ids in enumerate(hi):
    ids = ids.long()
    hit = ids[:, 0] >= 0
    node = ids[:, 0] * (1 << 32) + ids[:, 7]  # (first, last corner) identifies the node
    chg = torch.ones(pts, dtype=torch.bool, device=ids.device)
    chg[1:] = node[1:] != node[:-1]
    runs = int((chg & hit).sum())
    line = "  level %d (bottom-up): hits %d, node runs %d -> %d row-atomics" % (k, int(hit.sum()), runs, runs * 8)
    tot_runs += runs
    for W in (16, 64, 512, 4096, 32768, pts):
        n_win = (pts + W - 1) // W
        win = torch.arange(pts, device=ids.device) // W
        key = (win[:, None] * (1 << 40) + ids)[hit]  # (window, corner row)
        uniq = int(torch.unique(key.flatten()).numel())
        line += "; distinct corners per %s-point window: %d" % ("all" if W == pts else W, uniq)
    print(line)
print("  total node runs %d = %d row-atomics = %d fp32 atomics (at ~160 per ns: %.1f us of L2 atomic time)"
      % (tot_runs, tot_runs * 8, tot_runs * 64, tot_runs * 64 / 160e3))
