#!/usr/bin/env python
"""Same-box A/B of the single-launch plan of <= 4096-point batches (k_plan_unsorted) against the counting sort (kernel_variant 0x800 forces
it): the plan alone by device events, then the loops it sits in — Tier B eager (fused step + fused Adam, batch planned per call) and
Tier A (the drop-in's C++ nodes, backward on the calling thread) — interleaved, median of 7 x 100 iterations."""
import os, statistics, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import FeatureOctree, StepOptions, dp, fused_train_step, losses, optim, synth

os.environ["SHINE_TIER_A_EXT"] = "1"
torch.autograd.set_multithreading_enabled(False)
for kind in sys.argv[1:] or ["maicity", "kitti"]:
    wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=3)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
    cfg.ekional_loss_on = False
    g = torch.Generator(device="cuda").manual_seed(1)
    sigma = cfg.sigma_sigmoid
    for n in (1024, 4096):
        coord, label, weight = synth.draw_batch(wl.pool, n, g)
        res = {}
        for name, bits in (("counting sort", 0x800), ("k_plan_unsorted", 0)):
            FeatureOctree.DEBUG_VARIANT_BITS = bits
            for _ in range(20):
                dp.plan_batch(octree, coord)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(500):
                dp.plan_batch(octree, coord)
            e1.record()
            host = (time.perf_counter() - t0) / 500 * 1e6
            torch.cuda.synchronize()
            res[name] = "device %.1f us, host issue %.1f us" % (e0.elapsed_time(e1) / 500 * 1e3, host)
        print(kind, "N=%d" % n, "plan_batch alone:", res, "(n_buckets %d)" % octree._n_buckets, flush=True)

    o = StepOptions(sigma=sigma, ekional_loss_on=False)
    FeatureOctree.DEBUG_VARIANT_BITS = 0
    for n in (4096, 16384, 65536):  # the fused step itself on a node-ordered and on an unordered batch (device time by events)
        coord, label, weight = synth.draw_batch(wl.pool, n, g)
        sperm, sslots = dp.plan_batch(octree, coord, _debug_variant=0x800)
        uslots = torch.empty_like(sslots)
        uslots[sperm.long()] = sslots
        uperm = torch.arange(n, dtype=torch.int32, device="cuda")
        res = {}
        for name, (pp, ss) in (("node order", (sperm, sslots)), ("as given", (uperm, uslots))):
            for _ in range(10):
                fused_train_step(octree, dec, coord, label, weight, o, perm=pp, slots=ss)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                fused_train_step(octree, dec, coord, label, weight, o, perm=pp, slots=ss)
            e1.record()
            torch.cuda.synchronize()
            res[name] = "%.1f us" % (e0.elapsed_time(e1) / 200 * 1e3)
        print(kind, "N=%d" % n, "fused step (two launches, planned batch given):", res, flush=True)
    n = 4096
    adam_b = optim.setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
    opt_a = optim.setup_optimizer(cfg, list(octree.parameters()), list(dec.parameters()))

    def tier_b():
        coord, label, weight = synth.draw_batch(wl.pool, n, g)
        fused_train_step(octree, dec, coord, label, weight, o)
        adam_b.step(zero_grad=True)

    def tier_a():
        coord, sdf_label, weight = synth.draw_batch(wl.pool, n, g)
        feature = octree.query_feature(coord)
        sdf_pred = dec.sdf(feature)
        weight = torch.abs(weight)
        cur_loss = 0.
        cur_loss += losses.sdf_bce_loss(sdf_pred, sdf_label, sigma, weight, False, cfg.loss_reduction)
        opt_a.zero_grad(set_to_none=True)
        cur_loss.backward()
        opt_a.step()

    loops = {"tier B eager": tier_b, "tier A (C++ nodes, calling thread)": tier_a}
    times = {(k, b): [] for k in loops for b in (0x800, 0)}
    for b in (0x800, 0):
        FeatureOctree.DEBUG_VARIANT_BITS = b
        for fn in loops.values():
            for _ in range(20):
                fn()
    for rep in range(7):
        for b in (0x800, 0):
            FeatureOctree.DEBUG_VARIANT_BITS = b
            for k, fn in loops.items():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    fn()
                torch.cuda.synchronize()
                times[(k, b)].append((time.perf_counter() - t0) / 100 * 1e3)
    FeatureOctree.DEBUG_VARIANT_BITS = 0
    for k in loops:
        print(kind, "N=4096 BCE", k, "counting sort %.3f ms (min %.3f) | k_plan_unsorted %.3f ms (min %.3f)" % (
            statistics.median(times[(k, 0x800)]), min(times[(k, 0x800)]), statistics.median(times[(k, 0)]), min(times[(k, 0)])), flush=True)
