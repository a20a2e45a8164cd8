"""Per-frame cost of the octree growth path of incremental mapping (bench.py --workload ncd-incre, "update+ranks"): surface-point
selection, FeatureOctree.update (shine_tables_grow + the feature-side appends), node ranks.  Host-synchronised pieces, ms."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import FeatureOctree, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.make_config("ncd", device=dev, tree_level_feat=3)
frames = list(synth.make_frames(cfg, frames=15, beams=64, azimuths=450, seed=42, device=dev))
octree = FeatureOctree(cfg)
rows = []
for fi, (coord, label, weight) in enumerate(frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts = coord[weight > 0]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    octree.update(pts, incremental_on=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    octree._require_tables(with_ranks=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2))
med = np.median(np.array(rows[3:]), axis=0) * 1e3
print("surface-point selection %.3f ms, update %.3f ms, ranks %.3f ms (median of frames 3..14, %d surface points)" % (
    med[0], med[1], med[2], int(pts.shape[0])))
# inside update: the library call alone
import ctypes as C  # noqa: E402
from shine_mapping_amd import _lib  # noqa: E402

t = octree._tables
L = octree.featured_level_num
c = _lib.StepConfig()
c.n_levels, c.max_level = L, octree.max_level
fresh, added = (C.c_int64 * L)(), (C.c_int64 * L)()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(_lib.lib().shine_tables_grow(t.handle, C.byref(c), pts.data_ptr(), pts.shape[0], fresh, added,
                                            torch.cuda.current_stream().cuda_stream), "grow")
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("shine_tables_grow alone (no new nodes: the same frame again) %.3f ms" % (np.median(ts) * 1e3))
