#!/usr/bin/env python
"""Node-run statistics of a sorted draw on the synthetic MaiCity-like map, on the CPU (no GPU needed): how many node runs
(= feature-grad atomics of the fused step) fall into each 16-point tile, per level, and how unevenly they spread over the
contiguous tile ranges the kernels hand to a wave (8 tiles at 2^18 points) or a workgroup (64 tiles).

    python tools/run_stats.py [levels=4] [points=262144]

Approximates the planner's node order by the leaf-level Morton order of the samples (the same order for hits)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from shine_mapping_amd import synth
from shine_mapping_amd.feature_octree import morton_encode

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
cfg = synth.make_config("maicity", device="cpu", tree_level_feat=L)
cs, ws = [], []
for c, l, w in synth.make_frames(cfg, 60, 64, 450, 42, "cpu"):
    cs.append(c)
    ws.append(w)
c, w = torch.cat(cs), torch.cat(ws)
res = 2 ** cfg.tree_level_world
leaf = morton_encode(torch.floor(torch.clamp(res * (c + 1) / 2, 0, res - 1)).long().numpy())
surf = w.numpy() > 0
nodes = [np.unique(leaf[surf] >> (3 * (L - 1 - s))) for s in range(L)]
idx = np.sort(np.random.default_rng(0).integers(0, leaf.size, N))
lk = np.sort(leaf)[idx]
tiles = N // 16
total = np.zeros(tiles, np.int64)
for s in range(L):
    key = lk >> (3 * (L - 1 - s))
    hit = np.isin(key, nodes[s])
    slot = np.where(hit, key, -1)
    chg = np.ones(N, bool)
    chg[1:] = slot[1:] != slot[:-1]
    per_tile = (chg & hit).reshape(tiles, 16).sum(1)
    total += per_tile
    print("level %d (%6d nodes): hit fraction %.2f, %.2f runs per tile (max %d)" % (s, nodes[s].size, hit.mean(), per_tile.mean(), per_tile.max()))
for name, k in (("wave range (8 tiles)", 8), ("workgroup range (64 tiles)", 64)):
    g = total.reshape(-1, k).sum(1)
    print("%-28s runs: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d   (max / mean %.2f)" % (
        name, g.mean(), np.percentile(g, 50), np.percentile(g, 90), np.percentile(g, 99), g.max(), g.max() / g.mean()))
