#!/usr/bin/env python
"""The memory system's own time for the fused step's access stream (tools/ubench/random_rows.hip): gather of the 8 corner rows
of every (point, level), the run-merged scatter's atomics, and both at once — on the corner ids of a REAL sorted batch of the map.
    python tools/ubench/random_rows.py [kind=kitti_large] [frames=2800] [azimuths=300] [points=1048576]"""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shine_mapping_amd import synth
from shine_mapping_amd.sampler import SortedPool

kind = sys.argv[1] if len(sys.argv) > 1 else "kitti_large"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2800
az = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pts = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
lib = C.CDLL(os.path.join(ROOT, "tools", "ab", "librandom_rows.so"))
wl = synth.build_workload(kind, frames=frames, device="cuda", seed=42, tree_level_feat=3, azimuths=az)
octree = wl.octree
octree._require_tables(with_ranks=True)
sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
idx = sp.draw(pts)
coord = sp.coord[idx.long()].contiguous()  # visiting order
slots = sp.slots[idx.long()]               # [n, L]
hidx = octree.get_indices(coord)           # bottom-up list of [n, 8] int64
L = octree.featured_level_num
ids = torch.stack([hidx[L - 1 - s].to(torch.int32) for s in range(L)], 1).contiguous()  # [n, L, 8], level slot s top-down
lvl = torch.arange(L, dtype=torch.uint8, device="cuda").repeat(pts).contiguous()
g_ids, g_lvl = ids.view(-1, 8), lvl
# node runs per level, in stream order
first = torch.ones_like(slots, dtype=torch.bool)
first[1:] = slots[1:] != slots[:-1]
keep = (first & (slots >= 0)).view(-1)
s_ids, s_lvl = g_ids[keep].contiguous(), g_lvl[keep].contiguous()
feats = [p.detach() for p in octree.hier_features]
grads = [torch.zeros_like(p) for p in feats]
fp = (C.c_void_p * 4)(*([f.data_ptr() for f in feats] + [None] * (4 - L)))
gp = (C.c_void_p * 4)(*([g.data_ptr() for g in grads] + [None] * (4 - L)))
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
rows_g = int((g_ids >= 0).sum())
rows_s = int((s_ids >= 0).sum())
print("%s: %d points, %d (point, level) entries (%d rows gathered), %d node runs (%d rows = %d fp32 atomics), tables %.0f MB" % (
    kind, pts, g_ids.shape[0], rows_g, s_ids.shape[0], rows_s, rows_s * 8, sum(f.numel() * 4 for f in feats) / 1e6), flush=True)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(best)


for blocks in (1024, 2048, 4096, 8192):
    tg = timed(lambda: lib.rr_gather(fp, C.c_void_p(g_ids.data_ptr()), C.c_void_p(g_lvl.data_ptr()), C.c_longlong(g_ids.shape[0]),
                                     C.c_void_p(out.data_ptr()), blocks, C.c_void_p(st)))
    ts = timed(lambda: lib.rr_scatter(gp, C.c_void_p(s_ids.data_ptr()), C.c_void_p(s_lvl.data_ptr()), C.c_longlong(s_ids.shape[0]),
                                      blocks, C.c_void_p(st)))
    tb = timed(lambda: lib.rr_both(fp, gp, C.c_void_p(g_ids.data_ptr()), C.c_void_p(g_lvl.data_ptr()), C.c_longlong(g_ids.shape[0]),
                                   C.c_void_p(s_ids.data_ptr()), C.c_void_p(s_lvl.data_ptr()), C.c_longlong(s_ids.shape[0]),
                                   C.c_void_p(out.data_ptr()), blocks // 2, C.c_void_p(st)))
    rmw = {}
    if hasattr(lib, "rr_scatter_rmw"):
        for name, depth, mode in (("rmw x1", 1, 0), ("rmw x4", 4, 0), ("rmw x8", 8, 0), ("store x8", 8, 1)):
            rmw[name] = timed(lambda: lib.rr_scatter_rmw(gp, C.c_void_p(s_ids.data_ptr()), C.c_void_p(s_lvl.data_ptr()),
                                                         C.c_longlong(s_ids.shape[0]), blocks, depth, mode, C.c_void_p(st)))
        print("  %4d workgroups x 256, the scatter stream WITHOUT atomics (plain row read-modify-write, loads in flight per wave): %s" % (
            blocks, ", ".join("%s %.1f us" % kv for kv in rmw.items())), flush=True)
    print("  %4d workgroups x 256: gather %.1f us (%.2f TB/s of row bytes), scatter %.1f us (%.1f atomics/ns), both in one launch %.1f us" % (
        blocks, tg, rows_g * 32 / tg / 1e6, ts, rows_s * 8 / ts / 1e3, tb), flush=True)
