#!/usr/bin/env python
"""bench.py — trained SDF samples/s (fwd+bwd) of the fused SHINE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload maicity|kitti|kitti-large|ncd-incre]
                    [--points P] [--levels L] [--frames F] [--exchange dense|touched] ...

Workloads (synthetic, shine_mapping_amd/synth.py; BASELINE.json configs):
  maicity      config 2: MaiCity-like street, 2^18 points/iter, 4-level octree, BCE                     (default)
  kitti        config 3: KITTI-like 600 m two-turn polyline, 2^20 points/iter, L=3, BCE + eikonal
  kitti-large  the same on an 8.4 km serpentine whose feature tables (> 256 MiB) do not fit the Infinity Cache
  ncd-incre    config 4: incremental mapping (shine_incre.py:86-195), N=4096, 50 iterations per frame, regulariser,
               fused Adam, importance sweep, octree growth on the device, one HIP graph per iteration
A "step" (batch workloads) is one pass of the hot path over one batch resident in HBM: sorted draw from the
node-ordered pool (also clears the dense grads, i.e. opt.zero_grad) -> fused query + decode + loss + backward
(shine_batch.py:123-209 minus the optimiser) [-> gradient exchange under data parallelism].  For ncd-incre a step is
one frame.  With N>1 ranks (torch.distributed.run, one process per GPU, RCCL): ONE global sorted draw (same seed
everywhere), rank r takes the r-th contiguous slice of P points (weak scaling), global normalisers come from the common
draw, and the grads are exchanged dense (one flat all-reduce) or as touched rows only (--exchange).

One JSON line on rank 0: the driver's contract + `roofline` + `cpu_baseline` (DESIGN.md §5 explains every field).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0      # spec (6290 measured copy)
L2_PEAK_GBS = 34500.0      # aggregate L2
MFMA_F32_PEAK_TF = 157.3   # exact-fp32 MFMA = vector rate

WORKLOADS = {
    "maicity": dict(preset="maicity", points=1 << 18, levels=4, frames=60, azimuths=450),
    "kitti": dict(preset="kitti", points=1 << 20, levels=3, frames=120, azimuths=450),
    "kitti-large": dict(preset="kitti_large", points=1 << 20, levels=3, frames=2800, azimuths=300),
    "ncd-incre": dict(preset="ncd", points=4096, levels=3, frames=24, azimuths=900),
}


def algorithmic_bytes_per_point(levels: int, feat: int = 8) -> int:
    """24 + L*(40 + 2*8*F*4): batch in/out, one node record and 8 corner rows read + written per level."""
    return 24 + levels * (40 + 2 * 8 * feat * 4)


def pmc_record(workload, points, levels):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes (tools/collect_profiles.sh writes
    profiles/r02_pmc_<workload>_<points>_L<levels>.json; bench.py cannot run the profiler on itself, so the figures are
    only reported for a configuration they were measured on)."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_%s_%d_L%d.json" % (workload, points, levels))
    if not os.path.isfile(path):
        return None
    try:
        return json.load(open(path))
    except Exception:
        return None


def step_info(octree, cfg_eik, n):
    from shine_mapping_amd import _lib

    cfg = octree.step_config(eikonal_on=1 if cfg_eik else 0)
    out = (C.c_int64 * 8)()
    _lib.check(_lib.lib().shine_train_step_info(C.byref(cfg), n, out), "shine_train_step_info")
    return dict(workgroups=out[0], waves=out[1], tile_points=out[2], mfma_flop_per_tile=out[3], lds_bytes=out[4],
                useful_flop_per_point=out[5])


def cpu_baseline(wl, seconds=12.0, n=4096, max_iters=400):
    """The oracle port of the reference's CPU path (same per-point dict lookups, the same torch CPU op sequence, the
    reference's Adam groups) on a bounded sample of the same workload: whole iterations — query + decode + loss +
    backward + Adam step (shine_batch.py:115-210, the reference's timing(s)/total) — of N=4096 points (the reference's
    own batch size, config/*/..._batch.yaml `bs`) drawn from the same pool / octree, for ~`seconds` of CPU work."""
    from oracle import shine_oracle as so

    cfg = wl.cfg
    ocfg = so.make_config(tree_level_world=cfg.tree_level_world, tree_level_feat=cfg.tree_level_feat,
                          leaf_vox_size=cfg.leaf_vox_size, sigma_sigmoid_m=cfg.sigma_sigmoid_m,
                          ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e,
                          loss_reduction=cfg.loss_reduction)
    oct_ = so.OracleOctree(ocfg)
    for lvl, tab in enumerate(wl.octree.nodes_lookup_tables):
        oct_.node_table[lvl] = tab
    oct_.hier_features = [p.detach().cpu().clone().requires_grad_(True) for p in wl.octree.hier_features]
    mlp = so.OracleDecoder(ocfg)
    mlp.load_state_dict({k: v.cpu() for k, v in wl.decoder.state_dict().items()})
    opt = so.adam_param_groups(oct_, mlp, 0.01)
    host_cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(123)
    pool_n = wl.pool.sdf_label.shape[0]
    pdev = wl.pool.coord.device

    def one_iteration():
        idx = torch.randint(0, pool_n, (n,), generator=g).to(pdev)
        c, l, w = wl.pool.coord[idx].cpu(), wl.pool.sdf_label[idx].cpu(), wl.pool.weight[idx].cpu()
        t0 = time.perf_counter()
        so.train_step(oct_, mlp, c, l, w, ocfg)
        t1 = time.perf_counter()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return time.perf_counter() - t0, t1 - t0

    # torch's default (all host cores) is pathological for these small ops on a many-core host, so give the
    # CPU path its best thread count: calibrate on one iteration each, then time with the winner.
    best_t, best_threads = None, 1
    for threads in sorted({host_cores, min(host_cores, 32), min(host_cores, 8), 1}, reverse=True):
        torch.set_num_threads(threads)
        one_iteration()  # warm-up at this thread count
        dt = one_iteration()[0]
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, threads
    torch.set_num_threads(best_threads)
    done, t_used, t_nopt, it = 0, 0.0, 0.0, 0
    while t_used < seconds and it < max_iters:
        a, b = one_iteration()
        t_used += a
        t_nopt += b
        done += n
        it += 1
    return {
        "value": done / max(t_used, 1e-9), "unit": "samples/s", "cores": best_threads, "kind": "port",
        "host_cores": host_cores, "value_without_adam": done / max(t_nopt, 1e-9),
        "ms_per_iteration": t_used / max(it, 1) * 1e3,
        "sample": "%d whole iterations (query+decode+loss+backward+Adam step, the reference's timing(s)/total) of N=%d "
                  "(reference batch size) from the same pool/octree; oracle/shine_oracle.py train_step + "
                  "torch.optim.Adam(betas=(0.9,0.99), eps=1e-15) on the reference's groups, torch %s CPU, thread count "
                  "calibrated over {all,32,8,1}" % (it, n, torch.__version__),
    }


def gpu_iteration_n4096(wl, spool_seed, iters=300, n=4096):
    """The same iteration definition on the GPU at the same N: one captured HIP graph {sorted draw, fused step, fused
    dense Adam (clears the grads)} replayed (loop.GraphedIteration) — the like-for-like partner of cpu_baseline."""
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
    it = GraphedIteration(octree, decoder, spool, adam, opts, n)
    for _ in range(20):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    with torch.no_grad():  # leave the workload as it was
        for p, s in zip(params, saved):
            p.copy_(s)
    return {"n": n, "us_per_iteration": dt * 1e6, "samples_per_s": n / dt,
            "what": "sorted draw + fused step + fused dense Adam, one HIP graph per iteration"}


def run_incremental(args, dev):
    """BASELINE config 4 (shine_incre.py:86-195): per frame {update -> optimiser re-creation -> pool plan -> 50 x
    {sorted draw, fused step (sum reduction, touched rows), regulariser, fused Adam} as ONE replayed HIP graph ->
    importance sweep}.  A step = one frame."""
    import numpy as np

    from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, synth
    from shine_mapping_amd.incre_learning import cal_feature_importance
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    spec = WORKLOADS["ncd-incre"]
    bs = args.points or spec["points"]
    iters = args.iters
    n_frames = args.warmup + args.steps
    cfg = synth.make_config("ncd", device=dev, lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0,
                            tree_level_feat=args.levels or spec["levels"])
    frames = list(synth.make_frames(cfg, frames=n_frames, beams=64, azimuths=spec["azimuths"], seed=42, device=dev))
    torch.manual_seed(0)
    octree, dec = FeatureOctree(cfg), Decoder(cfg)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum")
    split = np.zeros((n_frames, 5))
    loss = None
    torch.cuda.synchronize()
    t_start = None
    for fi, (coord, label, weight) in enumerate(frames):
        if fi == args.warmup:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        t0 = time.perf_counter()
        octree.update(coord[weight > 0], incremental_on=True)
        octree._require_tables(with_ranks=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if fi == 20:  # shine_incre.py:100-104: the decoder is frozen after the first frames
            for p in dec.parameters():
                p.requires_grad_(False)
            opts.decoder_grad_on = False
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
        pool = SortedPool(octree, coord, label, weight, seed=fi)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        step = GraphedIteration(octree, dec, pool, opt, opts, bs, lambda_forget=cfg.lambda_forget,
                                unroll=args.unroll)  # = iteration 1
        loss = step.run(iters - 1)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
        cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, bs, 2, "sum")
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        split[fi] = (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)
    dt = time.perf_counter() - t_start
    med = np.median(split[args.warmup:], axis=0) * 1e3
    wl = type("WL", (), {})()
    wl.cfg, wl.octree, wl.decoder = cfg, octree, dec
    wl.pool = type("P", (), {"coord": frames[-1][0], "sdf_label": frames[-1][1], "weight": frames[-1][2]})()
    out = {
        "metric": "trained SDF samples/sec (fwd+bwd)", "value": args.steps * iters * bs / dt, "unit": "samples/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "ncd-like quad, incremental mode (shine_incre.py), N=%d, %d iterations/frame, sum reduction + "
                        "regulariser lambda=%g, %d-level octree, fused Adam, importance sweep, device octree growth; a "
                        "step = one frame" % (bs, iters, cfg.lambda_forget, cfg.tree_level_feat),
            "points_per_iter_per_gpu": bs, "levels": cfg.tree_level_feat, "frames": args.steps,
            "samples_per_frame": int(np.mean([f[0].shape[0] for f in frames])),
            "corner_rows": [int(p.shape[0]) for p in octree.hier_features], "parallelism": "dp1",
            "launch": "%d iterations per hipgraph replay (loop.GraphedIteration), re-captured per frame" % args.unroll,
        },
        "frames_per_s": args.steps / dt,
        "per_frame_ms_median": {"update+ranks": med[0], "optimiser+pool plan": med[1],
                                "%d iterations (incl. graph capture)" % iters: med[2], "importance sweep": med[3],
                                "total": med[4]},
        "us_per_iteration": med[2] / iters * 1e3,
        "final_loss": float(loss),
    }
    # roofline of the dominant kernel at this batch size (HIP events around back-to-back launches of the fused kernel)
    out["roofline"] = kernel_roofline("ncd-incre", octree, dec, cfg, pool, bs, None)
    if not args.no_cpu_baseline:
        cb = cpu_baseline(wl, n=bs)
        out["cpu_baseline"] = cb
        gpu_iter = med[2] / iters * 1e-3
        out["speedup_vs_cpu_baseline"] = (bs / gpu_iter) / cb["value"]
        out["like_for_like"] = {"n": bs, "gpu_samples_per_s_in_loop": bs / gpu_iter, "cpu_samples_per_s": cb["value"],
                                "note": "both sides: whole iterations incl. Adam at N=%d (the CPU side has no regulariser "
                                        "term: it would only make it slower)" % bs}
    print(json.dumps(out))


def kernel_roofline(workload, octree, decoder, cfg, spool, points, n_surf_fn, launch_graph=True):
    """HIP events on the launch stream around R back-to-back launches of the fused kernel ALONE (kernel_variant bit
    0x2000 skips the partial-sum reduction launch, so the bracket holds exactly what rocprofv3 reports for
    shine::k_step_v1), averaged per launch; plus every roof it can be held against."""
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
    achieved = points * bpp / t / 1e9
    tiles = (points + info["tile_points"] - 1) // info["tile_points"]
    issued_tf = tiles * info["mfma_flop_per_tile"] / t / 1e12
    useful_tf = points * info["useful_flop_per_point"] / t / 1e12
    rows = [int(p.shape[0]) for p in octree.hier_features]
    pmc = pmc_record(workload, points, levels)
    traffic = float(pmc["hbm_bytes_per_launch"]) if pmc and pmc.get("hbm_bytes_per_launch") else None
    hbm_meas = None if traffic is None else traffic / t / 1e9 / HBM_PEAK_GBS
    l2_frac = achieved / L2_PEAK_GBS
    mfma_util = pmc.get("mfma_util") if pmc else None
    # Exact-fp32 MFMA executes on the SIMD's fp32 FMA lanes and does NOT overlap VALU work of another wave
    # (tools/ubench/mfma_valu_overlap.hip, profiles/r02_ubench_mfma_valu_overlap.txt: 0.473 ms MFMA-only, 0.307 ms
    # VALU-only, 0.759 ms together), so the decoder's matrix work and the vector instructions share ONE datapath: its
    # occupancy (PMC) is the compute roof of this kernel, not the MFMA rate alone.
    dp_util = pmc.get("fp32_datapath_util") if pmc else None
    fracs = {"hbm": hbm_meas if hbm_meas is not None else 0.0,
             "mfma": dp_util if dp_util is not None else issued_tf / MFMA_F32_PEAK_TF}
    bound = max(fracs, key=fracs.get)
    return {
        # SURVEY.md §8(d) figure: algorithmic (no-reuse) bytes / kernel time against the HBM peak
        "bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_bytes_per_point": bpp,
        "frac_of_measured_copy_6290GBs": achieved / 6290.0,
        "compulsory_bytes": int(sum(rows) * 32 * 2 + 24 * points),
        # the honest roofs: real HBM bytes (PMC) against HBM peak; the algorithmic bytes as if all served by L2; the
        # matrix pipe (issued MFMA FLOP incl. padding, useful decoder FLOP, the PMC busy counter) and the fp32 datapath
        # the MFMAs share with the vector instructions (PMC: MFMA busy + VALU issue cycles over SIMD-cycles)
        "hbm_frac_measured": hbm_meas, "l2_frac": l2_frac,
        "mfma_issued_tflops": issued_tf, "mfma_issued_frac": issued_tf / MFMA_F32_PEAK_TF,
        "mfma_useful_frac": useful_tf / MFMA_F32_PEAK_TF, "mfma_util": mfma_util,
        "fp32_datapath_util": dp_util,
        "tcp_active_frac": pmc.get("tcp_active_frac") if pmc else None,
        "regime": "bound by the SIMD fp32 datapath (MFMA + VALU, which do not overlap) plus exposed gather / atomic "
                  "latency: hbm measured %s, L2 %.2f, MFMA issued %.2f, fp32 datapath %s; `bound` names the nearer of the "
                  "two contract roofs" % ("n/a" if hbm_meas is None else "%.2f" % hbm_meas, l2_frac,
                                          issued_tf / MFMA_F32_PEAK_TF, "n/a" if dp_util is None else "%.2f" % dp_util),
        "launch_geometry": {k: info[k] for k in ("workgroups", "waves", "tile_points", "lds_bytes")},
        "pmc_source": None if pmc is None else pmc.get("source"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="default 200 (ncd-incre: 12 frames)")
    ap.add_argument("--warmup", type=int, default=-1, help="default 20 (ncd-incre: 3 frames)")
    ap.add_argument("--workload", default="maicity", choices=sorted(WORKLOADS))
    ap.add_argument("--points", type=int, default=0, help="points per iteration per GPU (default: the workload's)")
    ap.add_argument("--levels", type=int, default=0, help="tree_level_feat (default: the workload's)")
    ap.add_argument("--frames", type=int, default=0, help="scans the synthetic map is built from")
    ap.add_argument("--iters", type=int, default=50, help="ncd-incre: iterations per frame (config iters)")
    ap.add_argument("--unroll", type=int, default=1,
                    help="ncd-incre: iterations captured per HIP graph (measured: 1 -> 4.70 ms, 7 -> 5.15 ms, 12 -> 5.62 ms per "
                         "frame of 50 iterations: the graph is re-captured every frame, and capturing 7x the nodes costs "
                         "more than 43 saved replays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--exchange", default="auto", choices=["auto", "dense", "touched"],
                    help="data-parallel gradient exchange: one flat all-reduce of the dense grads, or only the rows the "
                         "global batch touched (auto: touched when the dense bucket exceeds 64 MB)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the data-parallel code path even at world size 1")
    args = ap.parse_args()
    incre = args.workload == "ncd-incre"
    if args.steps <= 0:
        args.steps = 12 if incre else 200
    if args.warmup < 0:
        args.warmup = 3 if incre else 20

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)

    if incre:
        if world > 1:
            raise SystemExit("ncd-incre is a single-GPU workload (BASELINE.json config 4)")
        return run_incremental(args, dev)

    from shine_mapping_amd import StepOptions, fused_train_step, synth
    from shine_mapping_amd import dp as shine_dp
    from shine_mapping_amd.sampler import SortedPool

    spec = WORKLOADS[args.workload]
    levels = args.levels or spec["levels"]
    points = args.points or spec["points"]
    frames = args.frames or spec["frames"]
    wl = synth.build_workload(spec["preset"], frames=frames, device=dev, seed=42, tree_level_feat=levels,
                              azimuths=spec["azimuths"])
    cfg, octree, decoder, pool = wl.cfg, wl.octree, wl.decoder, wl.pool
    n_global = points * world
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction,
                       ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, n_global=n_global)
    params = list(octree.hier_features) + decoder.fused_params()
    for p in params:
        p.grad = torch.zeros_like(p)
    reducer = shine_dp.TouchedRowReducer(list(octree.hier_features), decoder.fused_params(), dist)
    exchange = args.exchange
    if exchange == "auto":
        exchange = "touched" if reducer.dense_bytes() > (64 << 20) else "dense"
    octree._require_tables(with_ranks=True)
    # ONE pool order and ONE random stream on every rank: the global draw is common knowledge
    # (canonical: the plan leaves the samples of one node in atomic-retirement order, which differs between processes)
    spool = SortedPool(octree, pool.coord, pool.sdf_label, pool.weight, seed=1000, canonical=use_dist)
    flags = shine_dp.mark_touched(octree, spool, spool.draw(8)) if (use_dist and exchange == "touched") else None
    if flags is not None:
        for f in flags:
            f.zero_()

    def step_body(i):
        """global sorted draw (+ clear grads in the same pass) -> fused step on this rank's slice (-> exchange)"""
        # this rank's contiguous slice of the ONE global sorted draw (same seed / draw count on every rank): only the
        # slice's indices are generated (shine_sample_sorted_slice), so the draw does not grow with the world size
        idx = spool.draw(points, zero=reducer.flat, n_global=n_global, slice_begin=rank * points)
        n_surf = None
        if opts.ekional_loss_on:  # global surface count: local count + an 8-byte all-reduce
            n_surf = (spool.weight[idx.long()] > 0).sum()
            if use_dist:
                reducer.all_reduce_scalar(n_surf)
        loss, pred, _ = fused_train_step(octree, decoder, None, None, None, opts, n_surf=n_surf, pool=spool, idx=idx)
        if use_dist:
            if exchange == "touched":
                shine_dp.mark_touched(octree, spool, idx, flags)  # this rank's rows ...
                reducer.or_reduce_flags(flags)                    # ... OR-ed into the global row set
                reducer.all_reduce_touched(flags)
            else:
                reducer.all_reduce_grads()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The loop body has no host sync and no allocation outside torch's allocator, so it is captured into HIP graphs and
    # replayed (launch-bound inner loops belong in hipGraphs) — a ring of graphs, one per random stream id, since a
    # replay would otherwise redraw the same batch.  With the dense exchange the RCCL all-reduce is captured with it; the
    # touched-row exchange reads a row count on the host and stays eager.
    launch = "eager"
    graphs, graph_loss = [], []
    ring = 8
    if not args.no_graph and not (use_dist and exchange == "touched"):
        try:
            for i in range(ring):
                step_body(i)  # warm caches / allocate workspaces / RCCL channels outside capture
            barrier()
            for i in range(ring):
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    graph_loss.append(step_body(i))
                graphs.append(g_)
            launch = "hipgraph" + (" (all-reduce captured)" if use_dist else "")
        except Exception as e:  # capture not possible on this stack: measure eagerly and say so
            print("graph capture failed (%s); falling back to eager launches" % e, file=sys.stderr)
            graphs, graph_loss, launch = [], [], "eager"
            torch.cuda.synchronize()

    def step(i):
        if graphs:
            graphs[i % len(graphs)].replay()
            return graph_loss[i % len(graphs)]
        return step_body(i)

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    roof = kernel_roofline(args.workload, octree, decoder, cfg, spool, points, None, launch_graph=not args.no_graph)

    # the reference's whole iteration (timing(s)/total, shine_batch.py:225): step + optimiser.  Fused Adam clears the
    # grads in the same pass.  Reported next to `value`, never instead of it.
    iter_ms = adam_ms = None
    if not use_dist:
        from shine_mapping_amd.optim import setup_optimizer

        cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
        for p in params:
            p.grad = torch.zeros_like(p)
        adam = setup_optimizer(cfg, list(octree.parameters()), decoder.fused_params())

        def iteration(i):
            ix = spool.draw(points)
            ns = (spool.weight[ix.long()] > 0).sum() if opts.ekional_loss_on else None
            fused_train_step(octree, decoder, None, None, None, opts, n_surf=ns, pool=spool, idx=ix)
            adam.step(zero_grad=True)

        for i in range(3):
            iteration(i)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for i in range(args.steps):
            iteration(i)
        torch.cuda.synchronize()
        iter_ms = (time.perf_counter() - ti) / args.steps * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            adam.step(zero_grad=True)
        e1.record()
        torch.cuda.synchronize()
        adam_ms = e0.elapsed_time(e1) / 10

    if rank == 0:
        rows = [int(p.shape[0]) for p in octree.hier_features]
        out = {
            "metric": "trained SDF samples/sec (fwd+bwd)", "value": points * world * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s: %s, batch mode, %d points/iter/GPU, %d-level octree (levels %d..%d), F=8, decoder "
                            "8-32-32-1, %s" % (args.workload,
                                               {"maicity": "MaiCity-like 100 m street canyon",
                                                "kitti": "KITTI-like 600 m polyline with two turns",
                                                "kitti-large": "KITTI-like 8.4 km serpentine (map larger than the "
                                                               "256 MiB Infinity Cache)"}[args.workload],
                                               points, levels, cfg.tree_level_world - levels + 1, cfg.tree_level_world,
                                               "BCE+eikonal" if cfg.ekional_loss_on else "BCE"),
                "points_per_iter_per_gpu": points, "levels": levels, "frames": frames,
                "pool_samples": int(pool.sdf_label.shape[0]), "corner_rows": rows,
                "feature_table_bytes": int(sum(rows) * 32),
                "batch_order": "sorted draw from the node-ordered pool (f-3); under DP one global draw, rank r takes the "
                               "r-th contiguous slice",
                "parallelism": "dp%d" % world, "launch": launch,
                "grad_exchange": None if not use_dist else (
                    "touched rows: %d rows, %.1f MB per step (dense bucket %.1f MB)" % (
                        reducer.last_rows, reducer.last_bytes / 1e6, reducer.dense_bytes() / 1e6)
                    if exchange == "touched" else "dense flat all-reduce, %.1f MB per step" % (reducer.dense_bytes() / 1e6)),
            },
            "roofline": roof,
            "final_loss": float(loss),
            "iteration_with_fused_adam": None if iter_ms is None else {
                "ms_per_iteration": iter_ms, "samples_per_s": points / (iter_ms * 1e-3), "launch": "eager",
                "dense_adam_ms": adam_ms,
                "what": "sorted draw + fused step + fused dense Adam (also clears grads); reference timing(s)/total"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            try:
                lf = gpu_iteration_n4096(wl, 77)
                out["like_for_like"] = {
                    "n": 4096, "gpu": lf, "cpu_samples_per_s": out["cpu_baseline"]["value"],
                    "speedup": lf["samples_per_s"] / out["cpu_baseline"]["value"],
                    "note": "same N (4096, the reference's batch size) and the same iteration definition (incl. Adam) on "
                            "both sides; `speedup_vs_cpu_baseline` divides the headline %d-point GPU step by this CPU figure"
                            % points}
            except Exception as e:
                out["like_for_like"] = {"error": str(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
