#!/usr/bin/env python
"""A/B two BUILDS of libshine_hip.so inside one process on one box (box-to-box variation on the pool is ~5 %, larger
than most single-change effects).  Usage:

    git stash / checkout the old kernel source;  python -m shine_mapping_amd.build --force
    cp shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_a.so          (then restore, rebuild)
    cp shine_mapping_amd/lib/libshine_hip.so tools/ab/lib_b.so
    python tools/ab_build.py tools/ab/lib_a.so tools/ab/lib_b.so

Both libraries are loaded; tables are created by the first one and used by both (the opaque handle is plain process
memory; only valid while include/shine_hip.h and shine_internal.hpp are unchanged between the two builds)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth, _lib
from shine_mapping_amd.sampler import SortedPool

paths = sys.argv[1:]
handles = []
for pth in paths:
    h = C.CDLL(os.path.abspath(pth))
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    handles.append(h)
_lib._lib = handles[0]

for kind, pts, lv in (("maicity", 1 << 18, 4), ("kitti", 1 << 20, 3)):
    wl = synth.build_workload(kind, frames=60, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    params = list(octree.hier_features) + dec.fused_params()
    for p in params:
        p.grad = torch.zeros_like(p)
    octree._require_tables(with_ranks=True)
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
    idx = sp.draw(pts)
    ns = (sp.weight[idx.long()] > 0).sum() if cfg.ekional_loss_on else None
    o = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, kernel_variant=0x2000)
    res = {}
    for rep in range(4):
        for name, h in zip(paths, handles):
            _lib._lib = h
            for _ in range(5):
                fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(os.path.basename(name), []).append(e0.elapsed_time(e1) / 20 * 1e3)
    _lib._lib = handles[0]
    print(kind, {k: ["%.1f" % v for v in vs] for k, vs in res.items()})
