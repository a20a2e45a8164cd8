#!/usr/bin/env python
"""The like-for-like iteration (N = 4096, Adam included: bench.gpu_iteration_n4096) on the bench's maps, with the exact
active-row Adam and with the dense sweep, in one process:  python tools/iter4096_bench.py [maicity kitti kitti-large]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from shine_mapping_amd import synth

for name in (sys.argv[1:] or ["maicity", "kitti-large"]):
    spec = bench.WORKLOADS[name]
    wl = synth.build_workload(spec["preset"], frames=spec["frames"], device="cuda", seed=42, tree_level_feat=spec["levels"],
                              azimuths=spec["azimuths"])
    wl.octree._require_tables(with_ranks=True)
    for active in (True, False, True):
        r = bench.gpu_iteration_n4096(wl, 77, iters=int(os.environ.get("ITERS", 300)), active_rows=active)
        print(name, [int(p.shape[0]) for p in wl.octree.hier_features], json.dumps({k: (round(v, 3) if isinstance(v, float) else v)
                                                                              for k, v in r.items() if k != "what"}), flush=True)
    del wl
    torch.cuda.empty_cache()
