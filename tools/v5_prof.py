#!/usr/bin/env python
"""Per-role cycle counters of the role-specialised fused step (shine_step_v5.hip built with -DSHINE_V5_PROF=1):

    python tools/mk_variant.py v5prof -DSHINE_V5_PROF=1 shine_step_v5.hip
    python tools/v5_prof.py tools/ab/lib_v5prof.so [maicity:4 kitti:3 ...]

Every wave reports [role, tiles, setup cycles, loop cycles, cycles spent polling a hand-off counter]: a role whose polling
share is small is the bottleneck of its pipeline, the others wait for it."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth, _lib
from shine_mapping_amd.sampler import SortedPool

lib = sys.argv[1]
cases = [a for a in sys.argv[2:]] or ["maicity:4", "kitti:3"]
h = C.CDLL(os.path.abspath(lib))
for name, (res, args) in _lib._SIGNATURES.items():
    fn = getattr(h, name)
    fn.restype, fn.argtypes = res, args
_lib._lib = _lib._check = h
PTS = {"maicity": 1 << 18, "kitti": 1 << 20}
for case in cases:
    kind, lv = case.split(":")
    lv = int(lv)
    pts = PTS[kind]
    wl = synth.build_workload(kind, frames=60, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    for p in list(octree.hier_features) + dec.fused_params():
        p.grad = torch.zeros_like(p)
    octree._require_tables(with_ranks=True)
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
    idx = sp.draw(pts)
    ns = (sp.weight[idx.long()] > 0).sum() if cfg.ekional_loss_on else None
    o = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, kernel_variant=0x2000 | 5)
    for _ in range(3):
        fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
    nw = 8192
    buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
    h.shine_debug_set_profile_buffer(buf.data_ptr())
    fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
    torch.cuda.synchronize()
    h.shine_debug_set_profile_buffer(None)
    prof = buf.view(nw, 8).cpu().double()
    used = prof[prof[:, 3] > 0]
    nd = int(used[:, 0].max()) - 1
    names = {r: "decoder %d" % r for r in range(nd)}
    names[nd], names[nd + 1] = "gather", "scatter"
    print("%s L%d, %d points: %d waves reported" % (kind, lv, pts, used.shape[0]))
    for r in sorted(names):
        u = used[used[:, 0] == r]
        if not u.shape[0]:
            continue
        print("  %-10s tiles/wave %5.1f | setup %7.0f | loop mean %8.0f max %8.0f | polling mean %8.0f (%.0f %%) max %8.0f | busy per tile %6.0f" % (
            names[r], float(u[:, 1].mean()), float(u[:, 2].mean()), float(u[:, 3].mean()), float(u[:, 3].max()),
            float(u[:, 4].mean()), 100 * float(u[:, 4].mean() / u[:, 3].mean()), float(u[:, 4].max()),
            float(((u[:, 3] - u[:, 4]) / u[:, 1].clamp_min(1)).mean())))
