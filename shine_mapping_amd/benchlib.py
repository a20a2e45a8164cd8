"""benchlib — the measured legs behind bench.py (the CLI and the printed record live there).

  timed_windows        THE timed region of a batch workload: R windows of exactly K steps between barrier + synchronize pairs
  run_batch            a batch-mode workload (BASELINE configs 2, 3 and `kitti-large`; config 5 under torch.distributed)
  run_incremental      BASELINE config 4 (shine_incre.py:86-195), Tier B loop + the Tier A loop body beside it
  run_dp_rank          rank 0's share of BASELINE config 5 on ONE GPU, reduced with the eight ranks' REAL messages
  kernel_roofline      HIP events around back-to-back launches of the fused kernel alone, held against its roofs
  gpu_iteration_n4096  the like-for-like partner of the CPU baseline (N = 4096, Adam included)

The CPU baseline is NOT here: it runs the oracle (test infrastructure), which only bench.py's `cpu_baseline` leg may touch;
the legs take it as the `cpu_baseline` callable.  Every leg returns the FULL record (a dict); bench.compact() cuts the
line the driver parses out of it.
"""
import ctypes as C
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# /opt/skills/guides/MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0      # spec (6290 measured copy)
L2_PEAK_GBS = 34500.0      # aggregate L2
MFMA_F32_PEAK_TF = 157.3   # exact-fp32 MFMA = the fp32 vector rate: 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
PEAK_CLOCK_GHZ = 2.4
N_SIMD = 1024
L2_ATOMICS_PER_NS = 160.0  # global_atomic_add_f32 lanes per ns, chip-wide, 8 rows x 32 B per wave instruction (measured:
                           # tools/ubench/atomics_rows.hip, 51-53 us for 131072 wave instructions whatever the address pattern)
MFMA16_CYCLES = 32.0       # v_mfma_f32_16x16x4_f32 issue interval per SIMD (guide, per-instruction constants)
# fp32-datapath cycles of one plain VALU wave-instruction.  The guide's SIMD-32 figure is 2; tools/ubench/mfma_valu_overlap
# (profiles/r03_ubench_calibration.txt) measures 5.0 for ONE wave per SIMD issuing independent v_fma_f32 (mode 5) and 2.52
# per instruction per SIMD with two waves (mode 6) — the best sustained rate seen; the packed v_pk_fma_f32 modes 1/4 that
# round 2 quoted as "VALU 2.3" are two FMAs per lane.  MFMA + VALU from different waves do not overlap (mode 7: 2.34 M
# cycles = 1.06 M MFMA + 1.28 M VALU).  The roof uses the best case, 2.5.
VALU_CYCLES = 2.5

WORKLOADS = {
    # SURVEY.md 8(d)'s scan recipe in full since round 5: sensor poses 1 m apart (100 on the street, 600 on the polyline) x 64 beams
    # x 1800 azimuths -> pools of 65 M / 378 M samples (rounds 1-4: 60 x 450 and 120 x 450 scans, pools of 9.6 M / 18.8 M; the same
    # box reads 3.10 -> 3.04 and 4.18 -> 3.95 G samples/s: denser rays find 12 % more of the map, and a draw from the larger pool
    # costs more — profiles/r05_bench_recipe_vs_thin.txt)
    "maicity": dict(preset="maicity", points=1 << 18, levels=4, frames=100, azimuths=1800),
    "kitti": dict(preset="kitti", points=1 << 20, levels=3, frames=600, azimuths=1800),
    "maicity-thin": dict(preset="maicity", points=1 << 18, levels=4, frames=60, azimuths=450),  # rounds 1-4's scans, for continuity
    "kitti-thin": dict(preset="kitti", points=1 << 20, levels=3, frames=120, azimuths=450),
    "kitti-large": dict(preset="kitti_large", points=1 << 20, levels=3, frames=2800, azimuths=300),
    "ncd-incre": dict(preset="ncd", points=4096, levels=3, frames=24, azimuths=900),
}


def algorithmic_bytes_per_point(levels: int, feat: int = 8) -> int:
    """24 + L*(40 + 2*8*F*4): batch in/out, one node record and 8 corner rows read + written per level."""
    return 24 + levels * (40 + 2 * 8 * feat * 4)


def pmc_record(workload, points, levels):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes (tools/collect_profiles.sh writes
    profiles/r03_pmc_<workload>_<points>_L<levels>.json; bench.py cannot run the profiler on itself).  A record is only
    used while it describes the code that runs: tools/pmc_to_json.py stamps the sha256 of the kernel's machine code
    (tools/kernel_hash.py) and the record is dropped — every PMC-derived field becomes null — when the loaded
    libshine_hip.so holds a different kernel.  -> (record or None, reason)"""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s_%d_L%d.json" % (workload, points, levels))))
    if not found:
        return None, "no counter file for this configuration"
    path = found[-1]  # (the latest round's collection)
    if not os.path.isfile(path):
        return None, "no counter file for this configuration"
    try:
        rec = json.load(open(path))
    except Exception as e:
        return None, "unreadable counter file (%s)" % e
    try:
        from kernel_hash import kernel_code_sha256
        from shine_mapping_amd import _lib

        want = rec.get("kernel_code_sha256")
        name = rec.get("kernel") or ""
        have = kernel_code_sha256(name, _lib.LIB_PATH) if name else None
    except Exception as e:
        return None, "kernel hash check failed (%s)" % e
    if not want or want != have:
        return None, "counter file describes another build of the kernel (sha256 %s..., loaded %s...)" % (
            str(want)[:12], str(have)[:12])
    return rec, "kernel code sha256 %s matches %s" % (want[:16], os.path.basename(path))


def step_info(octree, cfg_eik, n):
    from shine_mapping_amd import _lib

    cfg = octree.step_config(eikonal_on=1 if cfg_eik else 0)
    out = (C.c_int64 * 8)()
    _lib.check(_lib.lib().shine_train_step_info(C.byref(cfg), n, out), "shine_train_step_info")
    return dict(workgroups=out[0], waves=out[1], tile_points=out[2], mfma_flop_per_tile=out[3], lds_bytes=out[4],
                useful_flop_per_point=out[5])


def gpu_iteration_n4096(wl, spool_seed, iters=300, n=4096, active_rows=True):
    """The same iteration definition on the GPU at the same N: one captured HIP graph {fused step, {partial sums, Adam, grads
    cleared, next sorted draw}} replayed (loop.GraphedIteration) — the like-for-like partner of cpu_baseline.  Adam is torch's
    dense Adam in exact arithmetic: rows that have had no gradient since the optimiser was created are skipped unread
    (m = v = g = 0 leaves them bit for bit unchanged), so the time depends on how much of the map the iterations so far have
    touched — reported for the first and the last third of the window, with the fraction of rows still untouched."""
    from shine_mapping_amd import StepOptions
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    cfg, octree, decoder = wl.cfg, wl.octree, wl.decoder
    cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
    params = list(octree.hier_features) + decoder.fused_params()
    saved = [p.detach().clone() for p in params]
    for p in params:
        p.grad = torch.zeros_like(p)
    adam = setup_optimizer(cfg, list(octree.parameters()), decoder.fused_params())
    spool = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=spool_seed)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction, ekional_loss_on=cfg.ekional_loss_on,
                       weight_e=cfg.weight_e)
    unroll = 20  # iterations per HIP graph: the graph is built once and lives for hundreds of iterations here, so the ~16 us of idle
    # GPU at every graph boundary are folded away (a library-built graph's nodes cost nothing once instantiated)
    # This leg runs right after the CPU baseline (tens of seconds with an idle GPU): without a stretch of device work in front, its
    # first window reads the clock ramp (59 instead of 38 us per iteration on one box of the pool).  The stretch must not be
    # iterations of THIS loop — the active-row tail's time depends on how many it has taken — so it is plain device work.
    busy = torch.empty(2048, 2048, device=spool.coord.device).normal_()
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.06:
        for _ in range(20):
            busy = torch.tanh(busy @ busy * 1e-3)
        torch.cuda.synchronize()
    del busy
    it = GraphedIteration(octree, decoder, spool, adam, opts, n, unroll=unroll, active_rows=active_rows)
    it.run(20)
    torch.cuda.synchronize()
    third = max(unroll, iters // 3 // unroll * unroll)
    windows = []
    for _ in range(3):
        t0 = time.perf_counter()
        it.run(third)
        torch.cuda.synchronize()
        windows.append((time.perf_counter() - t0) / third)
    dt = sum(windows) / 3
    untouched = None
    if it.active_rows:
        untouched = float(sum(int((f[:-1] == 0).sum()) for f in it.touched)) / max(1, sum(f.numel() - 1 for f in it.touched))
    with torch.no_grad():  # leave the workload as it was
        for p, s in zip(params, saved):
            p.copy_(s)
    return {"n": n, "us_per_iteration": dt * 1e6, "samples_per_s": n / dt,
            "us_per_iteration_first_third": windows[0] * 1e6, "us_per_iteration_last_third": windows[2] * 1e6,
            "iterations_timed": 3 * third, "active_row_adam": bool(it.active_rows), "rows_never_touched_frac": untouched,
            "what": "fused step + {partial sums, Adam (exact, untouched rows skipped), grads cleared, next sorted draw}: two "
                    "launches per iteration, %d iterations per HIP graph" % unroll}


def run_incremental(args, dev, steps, warmup, cpu_baseline=None, cpu_seconds=12.0):
    """BASELINE config 4 (shine_incre.py:86-195): per frame {update -> optimiser re-creation -> pool plan -> 50 x
    {sorted draw, fused step (sum reduction, touched rows), regulariser, fused Adam} as ONE replayed HIP graph ->
    importance sweep}.  A step = one frame.  -> the bench record (dict)."""
    import numpy as np

    from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, synth
    from shine_mapping_amd.incre_learning import cal_feature_importance
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    spec = WORKLOADS["ncd-incre"]
    bs = (args.points if args.workload == "ncd-incre" else 0) or spec["points"]
    iters = args.iters
    n_frames = warmup + steps
    cfg = synth.make_config("ncd", device=dev, lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0,
                            tree_level_feat=(args.levels if args.workload == "ncd-incre" else 0) or spec["levels"])
    frames = list(synth.make_frames(cfg, frames=n_frames, beams=64, azimuths=spec["azimuths"], seed=42, device=dev))
    torch.manual_seed(0)
    octree, dec = FeatureOctree(cfg), Decoder(cfg)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum")
    split = np.zeros((n_frames, 5))
    loss = None
    torch.cuda.synchronize()
    t_start = None
    pool = step = opt = None
    # A full (generation-2) collection of the interpreter's heap — every module this process imported — lands on a fixed frame
    # (the allocation count decides) and costs 35-65 ms of a 3.4 ms frame: the objects alive now are moved out of the
    # collector's way (gc.freeze); the loop's own garbage is still collected.
    import gc

    gc.collect()
    gc.freeze()
    # The phases of a frame are delimited by events on the stream, so the host prepares the iterations (optimiser state, graph
    # binding) while the GPU is still growing the tree and planning the pool — as in the reference's loop, which has no
    # synchronisation between its phases either.  `split` = GPU time between the phase events; `host` = when the host was done
    # issuing a phase.
    # Default (round 4, second pass): NO synchronisation between frames either.  octree.update() is the one place where the host
    # reads the device (the counts of new nodes / corners); with FeatureOctree.enable_async_growth() the growth runs on a stream
    # of its own — it touches the hash tables only, which the queued iterations read through memoised slots — so the host binds
    # frame k + 1 while the device still trains frame k (two iteration graphs alternate: re-binding one waits for its own last
    # replay).  --sync-frames: one host synchronisation at the end of every frame (the round-3 form: a frame's latency).
    pipelined = not args.sync_frames
    if pipelined:
        octree.enable_async_growth()
    host = np.zeros((n_frames, 4))
    evs = []
    wall = np.zeros(n_frames)
    for fi, (coord, label, weight) in enumerate(frames):
        if fi == warmup:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        evs.append(ev)
        t0 = time.perf_counter()
        ev[0].record()
        if pipelined:  # (the frame's surface points come from static scan data: selected on the growth's stream, update()'s contract)
            with torch.cuda.stream(octree.growth_stream):
                surf = coord[weight > 0]
        else:
            surf = coord[weight > 0]
        octree.update(surf, incremental_on=True, ready=False if pipelined else None)  # (selected on the growth's own stream)
        octree._require_tables(with_ranks=True, probe=False)
        ev[1].record()
        t1 = time.perf_counter()
        if fi == 20:  # shine_incre.py:100-104: the decoder is frozen after the first frames
            for p in dec.parameters():
                p.requires_grad_(False)
            opts.decoder_grad_on = False
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
        pool = SortedPool(octree, coord, label, weight, seed=fi)
        ev[2].record()
        t2 = time.perf_counter()
        # frame 0: the constructor runs iteration 1 eagerly; later frames capture straight away and replay all of them
        step = GraphedIteration(octree, dec, pool, opt, opts, bs, lambda_forget=cfg.lambda_forget, unroll=args.unroll,
                                eager_first=fi == 0, graph_slot=fi % 2 if pipelined else 0)
        loss = step.run(iters - 1 if step.ran_eager else iters)
        ev[3].record()
        t3 = time.perf_counter()
        data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
        cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, bs, 2, "sum", pool=pool)  # (re-uses the frame's plan)
        ev[4].record()
        t4h = time.perf_counter()
        if not pipelined:
            torch.cuda.synchronize()
        wall[fi] = time.perf_counter() - t0
        host[fi] = (t1 - t0, t2 - t1, t3 - t2, t4h - t3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    gc.unfreeze()
    for fi, ev in enumerate(evs):
        # the frame's span: sequential form = the host's clock around the frame; pipelined = from the end of the previous frame's
        # last kernel to the end of this one's on the stream (the frame PERIOD: frames overlap on the host side)
        span = wall[fi] if (not pipelined or fi == 0) else evs[fi - 1][4].elapsed_time(ev[4]) * 1e-3
        split[fi] = tuple(ev[k].elapsed_time(ev[k + 1]) * 1e-3 for k in range(4)) + (span,)
    med = np.median(split[warmup:], axis=0) * 1e3
    wl = type("WL", (), {})()
    wl.cfg, wl.octree, wl.decoder = cfg, octree, dec
    wl.pool = type("P", (), {"coord": frames[-1][0], "sdf_label": frames[-1][1], "weight": frames[-1][2]})()
    out = {
        "metric": "trained SDF samples/sec (fwd+bwd)", "value": steps * iters * bs / dt, "unit": "samples/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "ncd-incre: ncd-like quad, incremental mode (shine_incre.py), N=%d, %d iterations/frame, sum "
                        "reduction + regulariser lambda=%g, %d-level octree, fused Adam, importance sweep, device octree "
                        "growth; a step = one frame" % (bs, iters, cfg.lambda_forget, cfg.tree_level_feat),
            "workload_short": "ncd-incre (BASELINE config 4): NCD-like quad, incremental mode, N=%d, %d iterations/frame, sum "
                              "reduction + regulariser, %d-level octree, fused Adam, importance sweep; a step = one frame" % (
                                  bs, iters, cfg.tree_level_feat),
            "points_per_iter_per_gpu": bs, "levels": cfg.tree_level_feat, "frames": steps,
            "samples_per_frame": int(np.mean([f[0].shape[0] for f in frames])),
            "corner_rows": [int(p.shape[0]) for p in octree.hier_features], "parallelism": "dp1",
            "launch": "%d iterations per hipgraph replay (loop.GraphedIteration): the graph is built by the library and its "
                      "kernel nodes are re-bound per frame (shine_iter_graph_*), no capture" % args.unroll,
        },
        "frames_per_s": steps / dt,
        "per_frame_ms_median": {"update+ranks": med[0], "optimiser+pool plan": med[1],
                                "%d iterations (incl. graph binding)" % iters: med[2], "importance sweep": med[3],
                                "total": med[4],
                                "note": ("phases: time between events on the stream; total: the frame period on the stream (no host "
                                         "synchronisation between frames: the host binds frame k + 1 while the device trains frame k)"
                                         if pipelined else
                                         "phases: time between events on the stream (the host runs ahead: one synchronisation "
                                         "per frame, at its end); total: host clock")},
        "frame_sync": not pipelined,
        "per_frame_host_issue_ms_median": dict(zip(("update+ranks", "optimiser+pool plan", "iterations", "importance sweep"),
                                                   [float(x) for x in np.median(host[warmup:], axis=0) * 1e3])),
        "us_per_iteration": med[2] / iters * 1e3,
        "per_frame_total_ms": [round(float(x) * 1e3, 3) for x in split[warmup:, 4]],
        "slowest_frame_split_ms": [round(float(x) * 1e3, 3) for x in split[warmup + int(np.argmax(split[warmup:, 4]))]],
        "iteration_graph": (lambda g: dict(zip(("commits", "builds"), g.stats())))(
            __import__("shine_mapping_amd.loop", fromlist=["IterationGraph"]).IterationGraph.shared(dev, args.unroll)),
        "final_loss": float(loss),
    }
    # roofline of the dominant kernel at this batch size (HIP events around back-to-back launches of the fused kernel)
    out["roofline"] = kernel_roofline("ncd-incre", octree, dec, cfg, pool, bs, None)
    # the same configuration through the UNCHANGED driver's names (Tier A: shine_incre.py's loop body verbatim on the drop-in's
    # classes, every launch issued by Python; tools/tier_a_bench.py) next to the fused loop above (Tier B)
    try:
        if args.no_tier_a:
            raise RuntimeError("skipped (--no-tier-a)")
        from tier_a_bench import tier_a_incremental

        n_a = min(len(frames), 8)
        out["tier_a"] = tier_a_incremental(dev, frames[:n_a], bs=bs, iters=iters, warmup=2, levels=cfg.tree_level_feat)
        out["tier_a"]["vs_tier_b_frames_per_s"] = out["tier_a"]["frames_per_s"] / out["frames_per_s"]
    except Exception as e:
        out["tier_a"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if cpu_baseline is not None:
        cb = cpu_baseline(wl, n=bs, seconds=cpu_seconds, regularize=True)
        out["cpu_baseline"] = cb
        gpu_iter = med[2] / iters * 1e-3
        out["like_for_like"] = {"n": bs, "gpu_samples_per_s_in_loop": bs / gpu_iter, "cpu_samples_per_s": cb["value"],
                                "speedup": (bs / gpu_iter) / cb["value"],
                                "note": "both sides: whole iterations incl. the regulariser term and Adam at N=%d" % bs}
    return out


def kernel_roofline(workload, octree, decoder, cfg, spool, points, n_surf_fn, launch_graph=True):
    """HIP events on the launch stream around R back-to-back launches of the fused kernel ALONE (kernel_variant bit
    0x2000 skips the partial-sum reduction launch, so the bracket holds exactly what rocprofv3 reports for the
    shine::k_step_* kernel), averaged per launch, held against the roof that binds it.

    `bound`/`achieved`/`peak`/`frac` are the "hbm" roof below; the "mfma" one is reported beside it under `datapath`:
      "mfma": the SIMD fp32 datapath.  Exact-fp32 MFMA runs on the same 64 FLOP/clk/SIMD lanes as the vector
              instructions and the two do not overlap (tools/ubench/mfma_valu_overlap.hip), so the compute roof of this
              kernel is the issued datapath work: (32 cycles x MFMAs + VALU_CYCLES x VALU instructions) per SIMD, expressed
              in FLOP at 64 FLOP/clk against the 157.3 TFLOP/s peak.  The VALU count needs the PMC record; without one
              only the MFMA share is counted (a lower bound).
      "hbm":  SURVEY.md §8(d)'s algorithmic (no-reuse) bytes against 8 TB/s — since round 6 ALWAYS the record's
              `bound` / `achieved` / `frac` (the contract's formula); it binds for maps beyond the Infinity Cache
              (kitti-large) and saturates near 1.0 for cache-resident ones (the table is served by L2 / Infinity Cache).
    """
    import copy

    from shine_mapping_amd import StepOptions, fused_train_step

    levels = cfg.tree_level_feat
    eik = bool(cfg.ekional_loss_on)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction, ekional_loss_on=eik,
                       weight_e=cfg.weight_e)
    kopts = copy.copy(opts)
    kopts.kernel_variant = 0x2000
    R = 10
    idx0 = spool.draw(points)
    ns0 = (spool.weight[idx0.long()] > 0).sum() if eik else None
    for p in list(octree.hier_features) + decoder.fused_params():
        if p.grad is None:
            p.grad = torch.zeros_like(p)

    def fused_only():
        for _ in range(R):
            fused_train_step(octree, decoder, None, None, None, kopts, n_surf=ns0, pool=spool, idx=idx0)

    fused_only()
    torch.cuda.synchronize()
    kg = None
    if launch_graph:
        try:
            kg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(kg):
                fused_only()
        except Exception:
            kg = None
            torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if kg is not None:
            kg.replay()
        else:
            fused_only()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / R)
    kernel_ms = sorted(times)[len(times) // 2]
    t = kernel_ms * 1e-3
    info = step_info(octree, eik, points)
    bpp = algorithmic_bytes_per_point(levels)
    alg_gbs = points * bpp / t / 1e9
    tiles = (points + info["tile_points"] - 1) // info["tile_points"]
    mfma_per_tile = info["mfma_flop_per_tile"] / 2048.0  # 16x16x4: 2048 FLOP each
    issued_tf = tiles * info["mfma_flop_per_tile"] / t / 1e12
    useful_tf = points * info["useful_flop_per_point"] / t / 1e12
    rows = [int(p.shape[0]) for p in octree.hier_features]
    table_bytes = sum(rows) * 32
    # the feature-grad atomics of this launch, counted from the batch itself: one 64-lane global_atomic_add_f32 per node run of
    # the ordered stream and level (a run = consecutive samples in the same node; misses issue none)
    sl = spool.slots[idx0.long()]
    hit = sl >= 0
    first = torch.ones_like(hit)
    first[1:] = sl[1:] != sl[:-1]
    node_runs = int((first & hit).sum())
    atomics_us = node_runs * 64 / (L2_ATOMICS_PER_NS * 1e3)
    pmc, pmc_note = pmc_record(workload, points, levels)
    ctr = pmc.get("counters_per_launch", {}) if pmc else {}
    traffic = float(pmc["hbm_bytes_per_launch"]) if pmc and pmc.get("hbm_bytes_per_launch") else None
    hbm_meas = None if traffic is None else traffic / t / 1e9 / HBM_PEAK_GBS
    # issued fp32-datapath work of one launch, in cycles per SIMD-lane group and as FLOP at 64 FLOP/clk
    n_mfma = tiles * mfma_per_tile
    n_valu = None
    if "SQ_INSTS_VALU" in ctr:  # SQ_INSTS_VALU counts the MFMAs too
        n_valu = max(float(ctr["SQ_INSTS_VALU"]) - float(ctr.get("SQ_INSTS_MFMA", n_mfma)), 0.0)
    dp_cycles = MFMA16_CYCLES * n_mfma + (VALU_CYCLES * n_valu if n_valu is not None else 0.0)
    dp_tf = dp_cycles * 64.0 / t / 1e12
    dp_frac = dp_tf / MFMA_F32_PEAK_TF
    clk = pmc.get("kernel_shader_cycles") if pmc else None
    in_cache = table_bytes * 2 < (200 << 20)  # features + grads inside the 256 MiB Infinity Cache
    # THE roofline of the record: SURVEY.md §8(d) — algorithmic (no-reuse) bytes of one launch / the launch's duration against
    # the 8 TB/s HBM peak.  For a cache-resident map the figure saturates (the table is served by L2 / Infinity Cache) and the
    # units that are actually busy are the ones listed beside it: `hbm` (PMC bytes), `datapath`, `l2_atomics`.
    roof = {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
            "frac_8d": alg_gbs / HBM_PEAK_GBS, "map_in_infinity_cache": bool(in_cache)}
    roof.update({
        "traffic": traffic, "kernel_ms": kernel_ms,
        "what": "SURVEY.md 8(d): N x (24 + L x (40 + 2 x 8 x F x 4)) algorithmic bytes / kernel time (HIP events around %d "
                "back-to-back launches of the fused kernel alone) against the 8 TB/s HBM peak" % R,
        "datapath_what": "issued fp32-datapath work (32 cycles x %.0f MFMA + %.1f cycles x %s VALU instructions per tile, 64 FLOP "
                         "per cycle and SIMD) / kernel time, against the fp32 MFMA = vector peak" % (
                             mfma_per_tile, VALU_CYCLES, "n/a" if n_valu is None else "%.0f" % (n_valu / tiles)),
        # SURVEY.md §8(d)'s no-reuse figure, always reported, never the bound of a cache-resident map
        "algorithmic": {"bytes_per_point": bpp, "GBps": alg_gbs, "frac_of_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                        "frac_of_measured_copy_6290GBs": alg_gbs / 6290.0,
                        "note": "no-reuse model; > 1 means the table is served from L1 / L2 / Infinity Cache"},
        "hbm": {"traffic_bytes": traffic, "frac": hbm_meas, "compulsory_bytes": int(sum(rows) * 32 * 2 + 24 * points),
                "table_bytes": int(table_bytes),
                "note": "PMC FETCH_SIZE x2 + WRITE_SIZE (guide §HBM; the x2 is calibrated for wide streams only, so the "
                        "true read traffic of the 16-B row gathers lies between x1 and x2)"},
        "datapath": {"mfma_per_tile": mfma_per_tile, "valu_per_tile": None if n_valu is None else n_valu / tiles,
                     "valu_cycles_per_instruction": VALU_CYCLES, "frac": dp_frac,
                     "frac_at_measured_clock": None if not clk else dp_cycles / (clk * N_SIMD),
                     "mfma_issued_tflops": issued_tf, "mfma_issued_frac": issued_tf / MFMA_F32_PEAK_TF,
                     "mfma_useful_frac": useful_tf / MFMA_F32_PEAK_TF, "mfma_busy_pmc": pmc.get("mfma_util") if pmc else None},
        "l2_atomics": {"node_runs": node_runs, "fp32_atomics": node_runs * 64, "rate_per_ns": L2_ATOMICS_PER_NS,
                       "floor_ms": atomics_us * 1e-3, "frac_of_kernel_time": atomics_us * 1e-3 / kernel_ms,
                       "note": "the launch's feature-grad atomics at the chip's measured rate for this shape (8 rows x 32 B per "
                               "wave instruction: tools/ubench/atomics_rows.hip, profiles/r04_ubench_atomics_rows.txt): the "
                               "time the L2's atomic units are busy, hidden under the kernel's other work or not "
                               "(profiles/r04_ab_experiments.txt block 10)"},
        "wave_cycle_split": None if not pmc else pmc.get("wave_cycle_split"),
        "launch_geometry": {k: info[k] for k in ("workgroups", "waves", "tile_points", "lds_bytes")},
        "pmc": {"used": pmc is not None, "note": pmc_note, "source": None if pmc is None else pmc.get("source")},
    })
    return roof

# --------------------------------------------------------------------------------------------- launch / distributed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def maybe_spawn(args, argv, script):
    """`--gpus N` (N > 1) without a torchrun environment: become the launcher — N ranks of this script under
    torch.distributed.run on 127.0.0.1, one per GPU — and exit with its return code.  Returns when this process is a
    rank (or N == 1)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if args.gpus > 1 and int(env_world) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d" % (
                args.gpus, env_world, args.gpus))
        return
    if args.gpus <= 1:
        return
    if not args.launch_check:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible — refusing to fall back to fewer ranks" % (
                args.gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(script)] + list(argv)
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    sys.exit(subprocess.call(cmd, env=env))


def init_ranks(args):
    """-> (dist or None, world, rank, local_rank, backend).  RCCL (`nccl`) unless SHINE_BENCH_BACKEND says otherwise
    (the CPU test of the launcher uses gloo)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if not use_dist:
        return None, 1, 0, 0, None
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    import torch.distributed as dist

    backend = os.environ.get("SHINE_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no GPU (visible devices: %d)" % (rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    if dist.get_world_size() != world:
        raise SystemExit("bench.py: process group reports %d ranks, expected %d" % (dist.get_world_size(), world))
    return dist, world, rank, local_rank, backend


def launch_check(args, dist, world, rank, backend):
    """--launch-check: prove the launcher and the rendezvous without touching a GPU (tests/test_bench_launch.py): every
    rank contributes 1 to an all-reduce; rank 0 prints the contract fields that depend on the launch."""
    t = torch.ones(1)
    if dist is not None:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(t.item()),
                          "world_size_reported": dist.get_world_size() if dist is not None else 1, "backend": backend,
                          "config": {"parallelism": "dp%d" % world}, "gpus_requested": args.gpus}))
    if dist is not None:
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------- batch workloads


def timed_windows(run, steps, repeats, barrier, dist, world, dev):
    """THE TIMED REGION.  `run(k)` issues exactly k steps of the hot path (HIP-graph replays or eager launches, one fresh batch
    per step; under data parallelism the exchange is inside the step).  R windows of exactly K = `steps` steps, each bracketed by
    barrier + torch.cuda.synchronize() on BOTH sides and scored by its SLOWEST rank; the line reports the MEDIAN window (one
    window of the driver's K = 20 is ~2 ms and says nothing about its own spread).
    -> (seconds of the median window, [seconds per window], {"max","min"} ms per step over the ranks or None)"""
    windows, rank_windows = [], []
    for _ in range(max(1, int(repeats))):
        barrier()                              # dist.barrier() + torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)                             # exactly K steps
        barrier()
        dt_local = time.perf_counter() - t0
        if dist is not None:                   # MAX over ranks
            t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
            all_t = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(all_t, t)
            per = [float(x) for x in all_t]
            windows.append(max(per))
            rank_windows.append(per)
        else:
            windows.append(dt_local)
    dt = statistics.median(windows)
    rank_ms = None
    if dist is not None:
        per = rank_windows[min(range(len(windows)), key=lambda i: abs(windows[i] - dt))]
        rank_ms = {"max": max(per) / steps * 1e3, "min": min(per) / steps * 1e3}
    return dt, windows, rank_ms



def run_batch(args, workload, dist, world, rank, dev, steps, warmup, cpu_baseline=None, with_like_for_like=True,
              with_iteration=True, cpu_seconds=12.0):
    """One batch-mode workload -> the bench record (dict on rank 0, None elsewhere)."""
    from shine_mapping_amd import StepOptions, fused_train_step, synth
    from shine_mapping_amd import dp as shine_dp
    from shine_mapping_amd.sampler import SortedPool

    use_dist = dist is not None
    own = workload == args.workload  # command-line overrides apply to the requested workload only
    spec = WORKLOADS[workload]
    levels = (args.levels if own else 0) or spec["levels"]
    points = (args.points if own else 0) or spec["points"]
    frames = (args.frames if own else 0) or spec["frames"]
    wl = synth.build_workload(spec["preset"], frames=frames, device=dev, seed=42, tree_level_feat=levels,
                              azimuths=spec["azimuths"])
    cfg, octree, decoder, pool = wl.cfg, wl.octree, wl.decoder, wl.pool
    n_global = points * world
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction,
                       ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, n_global=n_global)
    params = list(octree.hier_features) + decoder.fused_params()
    for p in params:
        p.grad = torch.zeros_like(p)
    octree._require_tables(with_ranks=True)
    # ONE pool order and ONE random stream on every rank: the global draw is common knowledge
    # (canonical: the plan leaves the samples of one node in atomic-retirement order, which differs between processes)
    torch.cuda.synchronize()
    t_plan = time.perf_counter()
    spool = SortedPool(octree, pool.coord, pool.sdf_label, pool.weight, seed=1000, canonical=use_dist)
    torch.cuda.synchronize()
    pool_plan_ms = (time.perf_counter() - t_plan) * 1e3
    feats, dec_params = list(octree.hier_features), decoder.fused_params()
    idx_buf = torch.empty(points, dtype=torch.int32, device=dev)
    surf_parts = spool.surf_parts_buffer(points) if opts.ekional_loss_on else None
    # steps per HIP graph: given, or (default) the largest divisor of the K timed steps up to 20 — the driver's K = 20 is then
    # ONE replay, and the ~9 us of idle GPU at a graph boundary is paid once per K steps instead of once per step
    U = int(args.graph_steps) if args.graph_steps > 0 else max(d for d in range(1, 21) if steps % d == 0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Gradient exchange under data parallelism (--exchange):
    #   dense    one flat all-reduce of the whole bucket
    #   gather   every rank moves the rows ITS slice touched into a fixed-size message, ONE all-gather, every rank adds all
    #            messages back in rank order (dp.RowGatherReducer): no host read, graph-capturable, fewer bytes on the links
    #            but pack / unpack launches on every rank; with --micro-batches M > 1 the slice runs as M fused steps and the
    #            (asynchronous) all-gather of micro-batch k overlaps the fused kernel of micro-batch k + 1
    #   touched  all-reduce of the union of all ranks' touched rows (row count read on the host: eager launches)
    # auto = MEASURED: dense, gather and gather with micro-batches are each built (a gather candidate must first reproduce the
    # dense all-reduce of one real step on every rank), run for a few steps on this node, and the fastest — by the slowest
    # rank's clock — is rebuilt and timed.  The record says which one ran and what the others took.
    def build_runner(kind, m):
        """-> dict(run(k) -> loss, launch, reducer, ...) for one exchange; None if a gather candidate fails its check"""
        for p in params:
            p.grad = torch.zeros_like(p)
        if kind == "gather":
            reducer = shine_dp.RowGatherReducer(feats, dec_params, dist, async_op=m > 1)
        else:
            reducer = shine_dp.TouchedRowReducer(feats, dec_params, dist)
        flags = None
        if use_dist and kind == "touched":
            flags = shine_dp.mark_touched(octree, spool, spool.draw(8))
            for f in flags:
                f.zero_()

        # The first pass of the NEXT step's draw (the block sums of its spacings: it depends on nothing but the sampler's stream
        # id) rides on this step's reduction launch (StepOptions.next_draw), so the draw in front of a step is ONE launch
        # (shine_sample_sorted_finish) instead of two.  The very first draw of a runner does both passes itself.
        rider = spool.next_draw(points, surf_parts=surf_parts, n_global=n_global) if points + 1 > 16 * 1024 else None
        opts_last = opts
        if rider is not None:
            import copy

            opts_last = copy.copy(opts)
            opts_last.next_draw = rider
        primed = [False]
        def run_micro(idx, n_surf):
            """the rank's slice as m contiguous micro-batches: fused step (marks its rows) -> pack + all-gather; then add back"""
            loss = None
            for k in range(m):
                a_, b_ = k * points // m, (k + 1) * points // m
                l_, _, _ = fused_train_step(octree, decoder, None, None, None, opts_last if k == m - 1 else opts, n_surf=n_surf,
                                            pool=spool, idx=idx[a_:b_], touched=reducer.flags)
                loss = l_ if loss is None else loss + l_
                reducer.exchange(finish=k == m - 1)
            return loss

        if use_dist and kind == "gather":  # one real step through the candidate == the dense all-reduce of the same grads
            ok = 1
            try:
                idx = spool.draw(points, zero=reducer.flat, n_global=n_global, slice_begin=rank * points)
                ns = None
                if opts.ekional_loss_on:
                    ns = (spool.weight[idx.long()] > 0).sum()
                    reducer.all_reduce_scalar(ns)
                fused_train_step(octree, decoder, None, None, None, opts, n_surf=ns, pool=spool, idx=idx)
                want = reducer.flat.clone()
                dist.all_reduce(want)
                reducer.flat.zero_()
                run_micro(idx, ns)
                torch.cuda.synchronize()
                err = float((reducer.flat - want).abs().max()) / max(float(want.abs().max()), 1e-30)
                if reducer.overflowed() or not err <= 1e-5:
                    ok = 0
            except Exception as e:  # an exchange that cannot run here must not take the measurement down with it
                print("rank %d: gather exchange check failed: %s" % (rank, e), file=sys.stderr)
                ok = 0
            t = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if not int(t):
                return None

        # One stream: draw -> fused step -> reduction (-> exchange).  Every step draws its own fresh batch (the sampler's
        # stream id lives in device memory and advances with every draw).  (Drawing the next batch on a forked graph branch
        # under the fused kernel was measured slower — profiles/r03_ab_experiments.txt block 9.)
        def step_body():
            """draw (its first pass also clears the gradient bucket: opt.zero_grad()) -> fused step on this rank's slice
            (-> exchange)"""
            # this rank's contiguous slice of the ONE global sorted draw (same seed / draw count on every rank): only the
            # slice's indices are generated (shine_sample_sorted_slice), so the draw does not grow with the world size
            idx = spool.draw(points, out=idx_buf, zero=reducer.flat, graph_safe=True, n_global=n_global,
                             slice_begin=rank * points, surf_parts=surf_parts, pass1_done=primed[0])
            primed[0] = rider is not None
            # eikonal: the surface count of the batch comes out of the draw as 64 partial counts which the step's kernels add
            # up (no launch of its own); data parallel: the global count = sum of the parts + an 8-byte all-reduce
            n_surf = surf_parts
            if surf_parts is not None and use_dist:
                n_surf = surf_parts.sum()
                reducer.all_reduce_scalar(n_surf)
            if use_dist and kind == "gather":
                return run_micro(idx, n_surf)
            loss, pred, _ = fused_train_step(octree, decoder, None, None, None, opts_last, n_surf=n_surf, pool=spool, idx=idx)
            if use_dist:
                if kind == "touched":
                    shine_dp.mark_touched(octree, spool, idx, flags)  # this rank's rows ...
                    reducer.or_reduce_flags(flags)                    # ... OR-ed into the global row set
                    reducer.all_reduce_touched(flags)
                else:
                    reducer.all_reduce_grads()
            return loss

        # The loop body has no host sync and no allocation outside torch's allocator, so it is captured into HIP graphs and
        # replayed (launch-bound inner loops belong in hipGraphs): one graph of `--graph-steps` U consecutive steps — a
        # replay costs ~9 us of idle GPU at its boundary whatever it holds, so K steps run as K // U replays of it plus K % U
        # replays of a one-step graph; every step in either graph is the full body above with its own draw.  The collectives
        # of the dense / gather exchange are captured with it; the touched-row exchange reads a row count on the host and
        # stays eager.
        launch = "eager"
        graph_u = graph_1 = None
        loss_u = loss_1 = None
        if not args.no_graph and not (use_dist and kind == "touched"):
            try:
                for _ in range(3):
                    step_body()  # warm caches / allocate workspaces / RCCL channels outside capture
                barrier()
                graph_1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_1):
                    loss_1 = step_body()
                if U > 1:
                    graph_u = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph_u):
                        for _ in range(U):
                            loss_u = step_body()
                launch = "hipgraph, %d step%s per replay, fresh batch per step%s%s" % (
                    U, "s" if U > 1 else "", " (collectives captured)" if use_dist else "",
                    ", first pass of the next draw on the step's reduction launch" if rider is not None else "")
            except Exception as e:  # capture not possible on this stack: measure eagerly and say so
                print("graph capture failed (%s); falling back to eager launches" % e, file=sys.stderr)
                graph_u = graph_1 = None
                launch = "eager"
                torch.cuda.synchronize()

        def run(k):
            """exactly k steps -> the last step's loss (also kept as run.last_loss)"""
            out = None
            if graph_1 is None:
                for _ in range(k):
                    out = step_body()
            else:
                q, r = divmod(k, U) if graph_u is not None else (0, k)
                for _ in range(q):
                    graph_u.replay()
                    out = loss_u
                for _ in range(r):
                    graph_1.replay()
                    out = loss_1
            if out is not None:
                run.last_loss = out
            return out

        run.last_loss = None

        return dict(run=run, launch=launch, reducer=reducer, kind=kind, micro=m, keep=(graph_u, graph_1, flags))

    def build_chain_runner():
        """One rank, batches of >= 16 K points (round 6): a step is TWO launches.  The step's reduction launch also draws the NEXT
        batch (pass 2 of draw i + 1, pass 1 of draw i + 2) and clears the NEXT step's gradient bucket (sampler.DrawChain,
        cfg->draw_rider) — the stand-alone draw launch of rounds 3-5 (7 us + a launch gap of an 86 us step) is gone.  Two gradient
        buckets alternate: step k accumulates into bucket k & 1, which stays intact — the step's result — until step k + 1's launch
        clears it.  A captured graph bakes each step's parity in; `run` replays the graph of U steps only where the host's parity
        says the device is (one graph of U steps and one single-step graph per starting parity)."""
        import copy

        for p in params:
            p.grad = torch.zeros_like(p)
        reducer = shine_dp.TouchedRowReducer(feats, dec_params, None)
        flat = (reducer.flat, torch.zeros_like(reducer.flat))
        nf = len(feats)

        def views_of(fl):
            out, off = [], 0
            for p in params:
                out.append(fl[off: off + p.numel()].view_as(p))
                off += p.numel()
            return out[:nf], out[nf:]

        views = (([p.grad for p in feats], [p.grad for p in dec_params]), views_of(flat[1]))
        eik = bool(opts.ekional_loss_on)
        chain = spool.draw_chain(points, idx_buf, buckets=flat, surf=eik)
        chain_opts = []
        for par in (0, 1):
            o = copy.copy(opts)
            o.draw_rider = chain.rider[par]
            chain_opts.append(o)
        chain.prime()

        def step_body(par):
            return fused_train_step(octree, decoder, None, None, None, chain_opts[par], n_surf=chain.surf_parts[par] if eik else None,
                                    pool=spool, idx=idx_buf, grad_buffers=views[par])[0]

        def eager(k):
            out = None
            for _ in range(k):
                out = step_body(chain.parity)
                chain.parity ^= 1
            return out

        launch, g1, gu, loss1, lossu = "eager", [None, None], [None, None], [None, None], [None, None]
        if not args.no_graph:
            try:
                eager(4)  # warm caches / allocate workspaces outside capture (an even number: parity back at 0)
                barrier()
                for par in (0, 1):
                    g1[par] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1[par]):
                        loss1[par] = step_body(par)
                if U > 1:  # one graph of U steps per starting parity (the driver's W = 5 warm-up steps leave the chain at parity 1)
                    for par in (0, 1):
                        gu[par] = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gu[par]):
                            for j in range(U):
                                lossu[par] = step_body((par + j) & 1)
                launch = ("hipgraph, %d step%s per replay, fresh batch per step; a step = 2 launches: the fused kernel and its "
                          "reduction, which also draws the next batch and clears the next step's bucket (two buckets alternate)" % (
                              U, "s" if U > 1 else ""))
            except Exception as e:
                print("graph capture failed (%s); falling back to eager launches" % e, file=sys.stderr)
                g1, gu, launch = [None, None], [None, None], "eager"
                torch.cuda.synchronize()

        def run(k):
            """exactly k steps -> the last step's loss (also kept as run.last_loss)"""
            out = None
            if g1[0] is None:
                out = eager(k)
            else:
                while k > 0:
                    if gu[chain.parity] is not None and k >= U:
                        gu[chain.parity].replay()
                        out, k = lossu[chain.parity], k - U
                        chain.parity ^= U & 1
                    else:
                        g1[chain.parity].replay()
                        out, k = loss1[chain.parity], k - 1
                        chain.parity ^= 1
            if out is not None:
                run.last_loss = out
            return out

        run.last_loss = None
        return dict(run=run, launch=launch, reducer=reducer, kind="dense", micro=1, keep=(gu, g1, chain, flat, views))

    exchange, micro, exchange_note, tuned = args.exchange, 1, None, None
    want_m = max(1, int(args.micro_batches))
    if not use_dist:
        exchange = "dense"  # (no exchange at all on one rank)
    if use_dist and exchange == "auto":
        tuned = {}
        t_tune = time.perf_counter()
        for kind, m in [("dense", 1), ("gather", 1)] + ([("gather", want_m)] if want_m > 1 else []):
            # (a bound on the tuning itself: the SCALE run times the whole process; every rank takes the same decision)
            over = torch.tensor([1.0 if (tuned and time.perf_counter() - t_tune > 45.0) else 0.0], device=dev)
            dist.all_reduce(over, op=dist.ReduceOp.MAX)
            if float(over) > 0:
                tuned["%s%s" % (kind, "" if m == 1 else " x%d micro-batches" % m)] = "skipped: tuning budget (45 s) spent"
                continue
            r_ = build_runner(kind, m)
            name = kind if m == 1 else "%s x%d micro-batches" % (kind, m)
            if r_ is None:
                tuned[name] = "failed its check against the dense all-reduce"
                continue
            r_["run"](2 * U)
            barrier()
            t0 = time.perf_counter()
            r_["run"](3 * U)
            barrier()
            t = torch.tensor([(time.perf_counter() - t0) / (3 * U)], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # every rank sees the same number -> the same choice
            tuned[name] = float(t) * 1e3
            del r_
            _release(dev)
        best = min((v, k) for k, v in tuned.items() if isinstance(v, float))[1]
        exchange, micro = ("gather", want_m if "x" in best else 1) if best.startswith("gather") else ("dense", 1)
    elif use_dist and exchange == "gather":
        micro = want_m
    # auto: while the draw is launch-bound.  Same box, rocprofv3 (profiles/r06_timeline_two_launch_step.txt): 2^18 draws —
    # reduction 5.2 + draw 6.9 us + a gap -> one launch of 8.8 us (step 86.8 -> 83.1 us); 2^20 draws — 5.3 + 24.0 -> 30.2 us and the
    # fused kernel 3 us slower on its two alternating 42 MB buckets (step 265.5 -> 269.8 us): the rider's sampler blocks are quarters
    # of the reduction kernel's 1024-thread blocks (16 waves at every barrier of the scan) and the draw is work-bound there.
    want_rider = {"on": True, "off": False}.get(args.draw_rider, points <= (1 << 18))
    chain_mode = not use_dist and points + 1 > 16 * 1024 and want_rider
    runner = build_chain_runner() if chain_mode else build_runner(exchange, micro)
    if runner is None:  # an explicitly requested gather exchange that does not reproduce the dense all-reduce here
        exchange_note = "gather exchange (%d micro-batches) failed its check against the dense all-reduce: dense used" % micro
        exchange, micro = "dense", 1
        runner = build_runner(exchange, micro)
    run, launch, reducer = runner["run"], runner["launch"], runner["reducer"]

    # Clock ramp: the driver's own invocation (--steps 20 --warmup 5) times ~2 ms after ~0.5 ms of warm-up — the GPU would
    # still be climbing out of its idle clocks and the line would not be the steady state the longer runs under profiles/
    # show.  A fixed stretch of the SAME replays runs first: not timed, and reported (`preheat_ms`).  Then W warm-up steps,
    # then exactly K timed steps.
    preheat_ms = 0.0
    if args.preheat_ms > 0:
        barrier()
        t_pre = time.perf_counter()
        run(2 * U)
        barrier()
        est = torch.tensor([(time.perf_counter() - t_pre) / (2 * U)], device=dev, dtype=torch.float64)
        if dist is not None:  # every rank must replay the same number of steps (the collectives are in the step)
            dist.all_reduce(est, op=dist.ReduceOp.MAX)
        n_pre = int(min(4000, max(1, args.preheat_ms * 1e-3 / max(float(est), 1e-6))) // U + 1) * U
        run(n_pre)
        barrier()
        preheat_ms = (time.perf_counter() - t_pre) * 1e3
    run(warmup)
    dt, windows, rank_ms = timed_windows(run, steps, args.repeats, barrier, dist, world, dev)
    loss = run.last_loss

    gather_overflow = None
    if use_dist and exchange == "gather":  # a message too small for a step's rows makes that step's grads incomplete: say so
        t = torch.tensor([1 if reducer.overflowed() else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_overflow = bool(int(t))
        if gather_overflow and rank == 0:
            print("WARNING: a rank touched more rows than a gather message holds during the timed steps", file=sys.stderr)

    roof = kernel_roofline(workload, octree, decoder, cfg, spool, points, None, launch_graph=not args.no_graph)

    # the reference's whole iteration (timing(s)/total, shine_batch.py:225): step + optimiser.  Fused Adam clears the
    # grads in the same pass.  Reported next to `value`, never instead of it.
    iter_ms = adam_ms = None
    if not use_dist and with_iteration:
        from shine_mapping_amd.optim import setup_optimizer

        cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
        for p in params:
            p.grad = torch.zeros_like(p)
        adam = setup_optimizer(cfg, list(octree.parameters()), decoder.fused_params())

        def iteration():
            ix = spool.draw(points)
            ns = (spool.weight[ix.long()] > 0).sum() if opts.ekional_loss_on else None
            fused_train_step(octree, decoder, None, None, None, opts, n_surf=ns, pool=spool, idx=ix)
            adam.step(zero_grad=True)

        for _ in range(3):
            iteration()
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for _ in range(steps):
            iteration()
        torch.cuda.synchronize()
        iter_ms = (time.perf_counter() - ti) / steps * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            adam.step(zero_grad=True)
        e1.record()
        torch.cuda.synchronize()
        adam_ms = e0.elapsed_time(e1) / 10

    if rank != 0:
        return None
    rows = [int(p.shape[0]) for p in octree.hier_features]
    out = {
        "metric": "trained SDF samples/sec (fwd+bwd)", "value": points * world * steps / dt,
        "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%s: %s, batch mode, %d points/iter/GPU, %d-level octree (levels %d..%d), F=8, decoder "
                        "8-32-32-1, %s; synthetic scans: %d poses x 64 beams x %d azimuths (%s)" % (workload,
                                           {"maicity": "MaiCity-like 100 m street canyon",
                                            "maicity-thin": "MaiCity-like 100 m street canyon",
                                            "kitti": "KITTI-like 600 m polyline with two turns",
                                            "kitti-thin": "KITTI-like 600 m polyline with two turns",
                                            "kitti-large": "KITTI-like 8.4 km serpentine (map larger than the "
                                                           "256 MiB Infinity Cache)"}[workload],
                                           points, levels, cfg.tree_level_world - levels + 1, cfg.tree_level_world,
                                           "BCE+eikonal" if cfg.ekional_loss_on else "BCE", frames, spec["azimuths"],
                                           "SURVEY.md 8(d)'s scan recipe: poses 1 m apart, 64 x 1800 rays each"
                                           if spec["azimuths"] == 1800 and not (args.frames and own) else
                                           "thinner than SURVEY.md 8(d)'s 64 x 1800 rays per pose 1 m apart: the map is the full "
                                           "one, the sample pool is smaller"),
            "workload_short": "%s (BASELINE config %s): %s map, batch mode, %d points/iter/GPU, %d-level octree, F=8, decoder "
                              "8-32-32-1, %s, f32; %d poses x 64 beams x %d azimuths" % (
                                  workload, {"maicity": "2", "kitti": "3"}.get(workload, "-"),
                                  {"maicity": "MaiCity-like 100 m street", "maicity-thin": "MaiCity-like 100 m street",
                                   "kitti": "KITTI-like 600 m polyline", "kitti-thin": "KITTI-like 600 m polyline",
                                   "kitti-large": "KITTI-like 8.4 km serpentine (tables > Infinity Cache)"}[workload],
                                  points, levels, "BCE+eikonal" if cfg.ekional_loss_on else "BCE", frames, spec["azimuths"]),
            "points_per_iter_per_gpu": points, "levels": levels, "frames": frames,
            "pool_samples": int(pool.sdf_label.shape[0]), "corner_rows": rows,
            "feature_table_bytes": int(sum(rows) * 32),
            "batch_order": "sorted draw from the node-ordered pool (f-3); under DP one global draw, rank r takes the "
                           "r-th contiguous slice",
            "lookup": "hoisted to the pool plan: the Morton-keyed hash probe of get_indices (model/feature_octree.py:199-218) "
                      "runs once per pool plan (pool and tree are static in batch mode) — every step reads a memoised 4-byte "
                      "hash slot per (sample, level) and the node's 8 corner ids (32 B)",
            "pool_plan_ms": pool_plan_ms,
            "window_ms": {"median": dt * 1e3, "min": min(windows) * 1e3, "max": max(windows) * 1e3,
                          "all": [w * 1e3 for w in windows],
                          "note": "%d windows of exactly %d steps; value / ms_per_step are the median window's" % (
                              len(windows), steps)},
            "parallelism": "dp%d" % world, "launch": launch,
            "world_size_reported": dist.get_world_size() if dist is not None else 1,
            "rank_ms_per_step": rank_ms,
            "grad_exchange": None if not use_dist else (
                "touched rows: %d rows, %.1f MB per step (dense bucket %.1f MB)" % (
                    reducer.last_rows, reducer.last_bytes / 1e6, reducer.dense_bytes() / 1e6)
                if exchange == "touched" else
                "own rows all-gather: %d micro-batch(es) per step, %.1f MB message per rank and micro-batch (capacity %d rows; "
                "dense bucket %.1f MB), checked against the dense all-reduce before timing%s" % (
                    micro, reducer.last_bytes / 1e6, reducer.capacity, reducer.dense_bytes() / 1e6,
                    ", all-gather of micro-batch k under the fused kernel of k + 1" if micro > 1 else "")
                if exchange == "gather" else "dense flat all-reduce, %.1f MB per step" % (reducer.dense_bytes() / 1e6)),
            "preheat_ms": preheat_ms,
            "grad_exchange_note": exchange_note,
            "grad_exchange_tuning_ms_per_step": tuned,
            "grad_exchange_overflow": gather_overflow,
        },
        "roofline": roof,
        "final_loss": float(loss),
        "iteration_with_fused_adam": None if iter_ms is None else {
            "ms_per_iteration": iter_ms, "samples_per_s": points / (iter_ms * 1e-3), "launch": "eager",
            "dense_adam_ms": adam_ms,
            "what": "sorted draw + fused step + fused dense Adam (also clears grads); reference timing(s)/total"},
    }
    if cpu_baseline is not None and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl, seconds=cpu_seconds)
        if with_like_for_like:
            try:
                lf = gpu_iteration_n4096(wl, 77)
                out["like_for_like"] = {
                    "n": 4096, "gpu": lf, "cpu_samples_per_s": out["cpu_baseline"]["value"],
                    "speedup": lf["samples_per_s"] / out["cpu_baseline"]["value"],
                    "note": "same N (4096, the reference's batch size) and the same iteration definition (incl. Adam) on "
                            "both sides — the GPU/CPU ratio to quote; `value` is the %d-point step without the optimiser"
                            % points}
            except Exception as e:
                out["like_for_like"] = {"error": str(e)}
    return out


XGMI_LINK_GBS = 153.0  # per direction and link, 7 links per GPU (/opt/skills/guides: xGMI point-to-point)


def rank_messages(octree, decoder, spool, opts, points, world, draw_no, reducer=None, keep_dense=False):
    """BASELINE config 5's exchange with REAL peers on ONE device: ranks 0..world-1 of ONE global sorted draw (`draw_no`) of
    points * world samples run back to back — each rank's fused step on ITS contiguous slice with the global normalisers
    (opts.n_global, the global surface count: shine_batch.py:174-185 takes ONE mean over the batch), marking the rows it touches,
    then ITS own-rows message (dp.RowGatherReducer.pack).  The capacity rule is the real one: measured on the first exchange as
    1.5 x the MAX over ranks + 1024 (dp._measure_capacity takes that maximum with an all-reduce; here the ranks' counts are read
    one after the other).  -> dict(reducer, messages: [world] int32 tensors, rows: [world] rows per message,
    flagged: [world] rows flagged, dense: [world] clones of the rank's dense bucket (keep_dense), n_surf)"""
    from shine_mapping_amd import dp as shine_dp
    from shine_mapping_amd import fused_train_step

    n_global = points * world
    assert int(opts.n_global) == n_global
    feats, dec_params = list(octree.hier_features), decoder.fused_params()
    red = reducer or shine_dp.RowGatherReducer(feats, dec_params, None)
    spool.draws = draw_no
    whole = spool.draw(n_global)
    n_surf = (spool.weight[whole.long()] > 0).sum() if opts.ekional_loss_on else None
    del whole

    def rank_step(r):
        red.flat.zero_()
        spool.draws = draw_no  # every rank is at the same draw count
        idx = spool.draw(points, n_global=n_global, slice_begin=r * points)
        fused_train_step(octree, decoder, None, None, None, opts, n_surf=n_surf, pool=spool, idx=idx, touched=red.flags)

    flagged = []
    for r in range(world):  # what every rank flags (the capacity measurement of a first exchange; also reported)
        rank_step(r)
        flagged.append(int(red._flags_flat.count_nonzero()))
        red._flags_flat.zero_()
        red._flags_flat[torch.tensor(red._keep, device=red._flags_flat.device)] = 1
    if red.capacity is None:
        red.capacity = red._round_cap(int(1.5 * max(flagged)) + 1024)
    msgs, rows, dense = [], [], []
    for r in range(world):
        rank_step(r)
        if keep_dense:
            dense.append(red.flat.clone())
        m = red.pack()
        msgs.append(m)
        rows.append(int(m[0]))
    red.flat.zero_()
    spool.draws = draw_no + 1
    return dict(reducer=red, messages=msgs, rows=rows, flagged=flagged, dense=dense, n_surf=n_surf)


def run_dp_rank(args, dev, steps=60, warmup=10):
    """BASELINE config 5 (KITTI-like, 2^22 points per iteration over 8 GPUs) as far as ONE GPU can measure it: rank 0's whole
    share of a step — its 2^19-point slice of the ONE global sorted draw of 2^22 (only the slice is generated), the global
    normalisers, the fused step marking its rows, then the own-rows exchange: pack, the all-gather's result = rank 0's fresh
    message + the REAL messages of ranks 1..7 (packed once from their own slices of one such draw: rank_messages), 8 unpack-adds
    — against the same slice without any exchange.  Before anything is timed, the eight messages of that draw are reduced and
    compared with the single-process step on the whole 2^22 batch.  Everything except the wire is measured; `scale_model` adds
    the wire from the xGMI link model and is labelled MODELLED.  -> dict"""
    from shine_mapping_amd import StepOptions, fused_train_step, synth
    from shine_mapping_amd import dp as shine_dp
    from shine_mapping_amd.sampler import SortedPool

    world, spec = 8, WORKLOADS["kitti"]
    points = (1 << 22) // world
    n_global = points * world
    wl = synth.build_workload(spec["preset"], frames=spec["frames"], device=dev, seed=42, tree_level_feat=spec["levels"],
                              azimuths=spec["azimuths"])
    cfg, octree, decoder, pool = wl.cfg, wl.octree, wl.decoder, wl.pool
    params = list(octree.hier_features) + decoder.fused_params()
    octree._require_tables(with_ranks=True)
    spool = SortedPool(octree, pool.coord, pool.sdf_label, pool.weight, seed=1000, canonical=True)
    feats, dec_params = list(octree.hier_features), decoder.fused_params()
    idx_buf = torch.empty(points, dtype=torch.int32, device=dev)
    surf_parts = spool.surf_parts_buffer(points)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction, ekional_loss_on=cfg.ekional_loss_on,
                       weight_e=cfg.weight_e, n_global=n_global)

    def timed(body):
        for _ in range(warmup):
            body()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            body()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    # the eight ranks' real messages of draw 3, reduced, against ONE process stepping through the whole 2^22 batch
    for p in params:
        p.grad = torch.zeros_like(p)
    rm = rank_messages(octree, decoder, spool, opts, points, world, draw_no=3)
    red = rm["reducer"]
    red.add_messages(torch.cat(rm["messages"]), world)
    reduced = red.flat.clone()
    red.flat.zero_()
    spool.draws = 3
    fused_train_step(octree, decoder, None, None, None, opts, n_surf=rm["n_surf"], pool=spool, idx=spool.draw(n_global))
    single = red.flat
    nf = red.n_rows * red.F
    err_feat = float((reduced[:nf] - single[:nf]).abs().max() / single[:nf].abs().max())
    err_dec = float((reduced[nf:] - single[nf:]).abs().max() / single[nf:].abs().max())
    overflow_real = red.overflowed()
    peers = torch.cat(rm["messages"][1:]).clone()
    rows_ = list(rm["rows"])
    capacity = red.capacity
    del rm, reduced, single

    res = {}
    for kind in ("none", "gather"):
        for p in params:
            p.grad = torch.zeros_like(p)
        reducer = shine_dp.RowGatherReducer(feats, dec_params, None, synthetic_world=world if kind == "gather" else 0,
                                            capacity_rows=capacity)
        reducer.peer_messages = peers if kind == "gather" else None

        def body():
            idx = spool.draw(points, out=idx_buf, zero=reducer.flat, graph_safe=True, n_global=n_global, slice_begin=0,
                             surf_parts=surf_parts)
            n_surf = surf_parts.sum() * world  # (stands in for the 8-byte all-reduce of the ranks' counts)
            fused_train_step(octree, decoder, None, None, None, opts, n_surf=n_surf, pool=spool, idx=idx,
                             touched=reducer.flags if kind == "gather" else None)
            if kind == "gather":
                reducer.exchange()

        res[kind] = timed(body)
        if kind == "gather":
            msg_bytes, cap, dense = reducer.last_bytes, reducer.capacity, reducer.dense_bytes()
            overflow = reducer.overflowed() or overflow_real
    # the dense alternative's local part: nothing beyond the step itself (the bucket is cleared by the draw; the reduction is the
    # collective's own kernels), so its measured share is res["none"]
    t_none, t_gather = res["none"] * 1e-3, res["gather"] * 1e-3
    link = XGMI_LINK_GBS * 1e9
    wire = {
        "gather_direct_s": msg_bytes / link,                      # every peer's message over its own link, in parallel
        "gather_ring_s": (world - 1) * msg_bytes / link,          # one ring: 7 hops of one message each
        "dense_direct_s": 2.0 * (dense / world) / link,           # reduce-scatter + all-gather over all 7 links at once
        "dense_ring_s": 2.0 * (world - 1) / world * dense / link, # single ring
    }
    model = {}
    for name, local, w in (("gather, direct all-gather", t_gather, wire["gather_direct_s"]),
                           ("gather, ring", t_gather, wire["gather_ring_s"]),
                           ("dense, direct reduce-scatter + all-gather", t_none, wire["dense_direct_s"]),
                           ("dense, single ring", t_none, wire["dense_ring_s"])):
        t = local + w  # (no overlap assumed: the exchange follows the step)
        model[name] = {"ms_per_step": t * 1e3, "samples_per_s_8_gpus": n_global / t, "weak_scaling_efficiency_vs_1_gpu": t_none / t}
    return {
        "what": "rank 0's share of BASELINE config 5 on one GPU: slice of 2^19 of ONE global sorted draw of 2^22 points, KITTI-like "
                "map, BCE + eikonal with global normalisers; 'gather' adds the own-rows exchange with the REAL messages of ranks "
                "1..7 (pack, rank 0's fresh message + 7 stored real ones, 8 unpack-adds) — everything except the wire",
        "points_per_rank": points, "n_global": n_global, "world": world,
        "ms_per_step_measured": {"no exchange": res["none"], "own-rows exchange, 8 real messages": res["gather"]},
        "exchange_local_cost_ms": res["gather"] - res["none"],
        "message_rows_per_rank": {"min": min(rows_), "max": max(rows_), "all": rows_},
        "reduced_vs_single_process_max_rel_err": {"feature_grads": err_feat, "decoder_grads": err_dec},
        "message_bytes_per_rank": int(msg_bytes), "message_capacity_rows": int(cap), "dense_bucket_bytes": int(dense),
        "message_overflow": bool(overflow),
        "scale_model": {"MODELLED": True, "link_GBps": XGMI_LINK_GBS, "wire_s": wire, "by_exchange": model,
                        "note": "measured local time + message bytes / link bandwidth, no overlap, no collective launch latency; "
                                "no multi-GPU node was available to any round: these are not measurements"},
    }


def _release(dev):
    import gc

    gc.collect()
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
