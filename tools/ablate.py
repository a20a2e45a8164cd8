#!/usr/bin/env python
"""Ablation timing of the fused step's phases (debug bits of kernel_variant, see shine_step_v1.hip).

    python tools/ablate.py [--points P] [--levels L] [--frames F]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, dp, fused_train_step, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=1 << 18)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--workload", default="maicity")
args = ap.parse_args()

wl = synth.build_workload(args.workload, frames=args.frames, device="cuda", seed=42, tree_level_feat=args.levels)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
g = torch.Generator(device="cuda").manual_seed(1)
batches = [synth.draw_batch(wl.pool, args.points, g) for _ in range(4)]
params = list(octree.hier_features) + dec.fused_params()
red = dp.GradReducer(params)
for p in params:
    p.grad = torch.zeros_like(p)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


perms = [dp.morton_order(octree, b[0]) for b in batches]
plans = [dp.plan_batch(octree, b[0]) for b in batches]
state = {"i": 0}


def step(variant, sorted_=True, planned=True):
    i = state["i"] = (state["i"] + 1) % len(batches)
    c, l, w = batches[i]
    opts = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e,
                       kernel_variant=variant)
    if not sorted_:
        fused_train_step(octree, dec, c, l, w, opts)
    elif planned:
        fused_train_step(octree, dec, c, l, w, opts, perm=plans[i][0], slots=plans[i][1])
    else:
        fused_train_step(octree, dec, c, l, w, opts, perm=perms[i])


print("points %d levels %d rows %s" % (args.points, args.levels, [int(p.shape[0]) for p in octree.hier_features]))
print("radix sort only      : %8.1f us" % timeit(lambda: dp.morton_order(octree, batches[0][0])))
print("plan (node sort) only: %8.1f us" % timeit(lambda: dp.plan_batch(octree, batches[0][0])))
for nm, v in (("plan: no count atomics", 0x100), ("plan: no probes", 0x200), ("plan: no slot stores", 0x400),
              ("plan: none of the three", 0x700)):
    print("%-21s: %8.1f us" % (nm, timeit(lambda: dp.plan_batch(octree, batches[0][0], _debug_variant=v))))
print("plan + fused zero    : %8.1f us" % timeit(lambda: dp.plan_batch(octree, batches[0][0], zero=red.flat)))
print("radix-ordered step   : %8.1f us" % timeit(lambda: step(0, planned=False)))
print("zero grads           : %8.1f us" % timeit(red.zero_grads))
rows = [("full", 0), ("unsorted input", 0), ("no atomics", 0x100), ("no weight-grad phase", 0x200),
        ("no scatter phase", 0x400), ("no row gathers", 0x800), ("no probe (all miss)", 0x1000),
        ("no scatter+wgrad", 0x600), ("no scatter+wgrad+gather", 0xE00), ("nothing but decoder", 0x1E00)]
for name, v in rows:
    t = timeit(lambda: step(v, sorted_=(name != "unsorted input")))
    print("%-22s: %8.1f us" % (name, t))


# ---- in-kernel phase cycle counters (s_memtime), one full step
import ctypes
from shine_mapping_amd import _lib
lib = _lib.lib()
nw = 4096
buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
lib.shine_debug_set_profile_buffer(buf.data_ptr())
for variant in ([int(v, 0) for v in os.environ['SHINE_PROF_VARIANTS'].split(',')] if os.environ.get('SHINE_PROF_VARIANTS') else (0, 0x1E00)):
    buf.zero_()
    step(variant)
    torch.cuda.synchronize()
    prof = buf.view(nw, 8).cpu().double()
    used = prof[prof.sum(1) > 0]
    names = ["setup", "query", "dec fwd", "loss+dec bwd", "scatter", "wgrad", "flush", "block wait"]
    print("variant 0x%x: %d waves; mean cycles per wave (max):" % (variant, used.shape[0]))
    for k, nme in enumerate(names):
        print("   %-14s %10.0f  (%10.0f)" % (nme, float(used[:, k].mean()), float(used[:, k].max())))
    print("   %-14s %10.0f" % ("total", float(used.sum(1).mean())))
lib.shine_debug_set_profile_buffer(None)
