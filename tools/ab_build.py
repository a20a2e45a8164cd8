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

# AB_VARIANTS="0,3": every library is timed once per kernel_variant (low byte; 0 = the library's own choice);
# "lib.so@5,6" gives one library its own list (e.g. the far / near build of the fused step, whatever the table size)
variants = [int(v, 0) for v in os.environ.get("AB_VARIANTS", "0").split(",")]
lib_paths, paths, handles = [], [], []
for arg in sys.argv[1:]:
    pth, _, own = arg.partition("@")
    lib_paths.append(pth)
    h = C.CDLL(os.path.abspath(pth))
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(h, name, None)  # (an older build may lack the newest entry points)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    for v in ([int(x, 0) for x in own.split(",")] if own else variants):
        paths.append("%s:%d" % (pth, v))
        handles.append(h)
_lib._lib = handles[0]
kvar = {name: int(name.rsplit(":", 1)[1]) for name in paths}

CASES = (("maicity", 1 << 18, 4), ("kitti", 1 << 20, 3), ("maicity", 1 << 18, 3), ("kitti", 1 << 20, 4))
only = os.environ.get("AB_ONLY")  # e.g. "maicity:4,kitti:3"
if only:
    want = {tuple(x.split(":")) for x in only.split(",")}
    CASES = tuple(c for c in CASES if (c[0], str(c[2])) in want) or \
        tuple((k, 1 << 20, int(lv)) for k, lv in want)  # (e.g. AB_ONLY=kitti_large:3 AB_FRAMES=2800 AB_AZIMUTHS=300)
else:
    CASES = CASES[:2]
if os.environ.get("AB_POINTS"):  # e.g. AB_POINTS=4096: the reference's own batch size (every shipped yaml)
    CASES = tuple((k, int(os.environ["AB_POINTS"]), lv) for k, _, lv in CASES)
for kind, pts, lv in CASES:
    wl = synth.build_workload(kind, frames=int(os.environ.get("AB_FRAMES", 60)), device="cuda", seed=42, tree_level_feat=lv,
                              azimuths=int(os.environ.get("AB_AZIMUTHS", 450)))
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
            _lib._lib = _lib._check = h
            o.kernel_variant = 0x2000 | kvar[name]
            for _ in range(5):
                fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(os.path.basename(name), []).append(e0.elapsed_time(e1) / 20 * 1e3)
    # one full step (reduction included) per library: loss and gradients must agree with the first library's
    o1 = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e)
    ref = None
    for name, h in zip(paths, handles):
        _lib._lib = _lib._check = h
        o1.kernel_variant = kvar[name]
        for p in params:
            p.grad.zero_()
        loss, pred, _ = fused_train_step(octree, dec, None, None, None, o1, n_surf=ns, pool=sp, idx=idx)
        torch.cuda.synchronize()
        cur = [float(loss)] + [p.grad.double().clone() for p in params] + [pred.double().clone()]
        if ref is None:
            ref = cur
        else:
            errs = [float((c - r).abs().max() / r.abs().max().clamp_min(1e-30)) for c, r in zip(cur[1:], ref[1:])]
            print("  parity %-18s loss %.9g vs %.9g, max rel grad/pred err %.2e" % (os.path.basename(name), cur[0], ref[0], max(errs)))
    if os.environ.get("AB_PROF"):  # in-kernel phase cycle counters of every library's PROF instantiation
        nw = 8192
        buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
        names = ["setup", "query", "dec fwd", "loss+dec bwd", "scatter", "wgrad", "flush", "block wait"]
        for name, h in zip(paths, handles):
            _lib._lib = _lib._check = h
            o.kernel_variant = 0x2000 | kvar[name]
            h.shine_debug_set_profile_buffer(buf.data_ptr())
            buf.zero_()
            fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)
            torch.cuda.synchronize()
            h.shine_debug_set_profile_buffer(None)
            prof = buf.view(nw, 8).cpu().double()
            if os.environ.get("AB_PROF_DUMP"):
                slots_b = sp.slots[idx.long()].cpu()  # [n, L] hash slot per level of the drawn batch, visiting order
                torch.save({"prof": prof, "slots": slots_b, "lib": name, "kind": kind, "levels": lv},
                           os.path.join(os.environ["AB_PROF_DUMP"], "prof_%s_%s_L%d.pt" % (os.path.basename(name), kind, lv)))
            used = prof[prof.sum(1) > 0]
            if used.shape[0] == 0:
                print('  prof %-16s (this kernel has no instrumented instantiation)' % os.path.basename(name))
                continue
            tot = used.sum(1)
            print("  prof %-16s %d waves, cycles per wave mean (max): " % (os.path.basename(name), used.shape[0]) +
                  ", ".join("%s %.0f (%.0f)" % (nm, float(used[:, k].mean()), float(used[:, k].max())) for k, nm in enumerate(names)) +
                  "; total %.0f (min %.0f max %.0f); work excl. flush/wait: mean %.0f max %.0f" % (
                      float(tot.mean()), float(tot.min()), float(tot.max()),
                      float(used[:, :6].sum(1).mean()), float(used[:, :6].sum(1).max())))
    _lib._lib = handles[0]
    print(kind, lv, {k: ["%.1f" % v for v in vs] for k, vs in res.items()})
