#!/usr/bin/env python
"""bench.py — trained SDF samples/s (fwd+bwd) of the fused SHINE hot path on MI355X.  CLI + the printed record; the measured
legs are shine_mapping_amd/benchlib.py (the timed region: benchlib.timed_windows).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload maicity|kitti|kitti-large|ncd-incre]
                    [--points P] [--levels L] [--frames F] [--exchange gather|dense|touched] ...

Workloads (synthetic, shine_mapping_amd/synth.py; BASELINE.json configs):
  maicity      config 2: MaiCity-like street, 2^18 points/iter, 4-level octree, BCE                     (default, THE headline)
  kitti        config 3: KITTI-like 600 m two-turn polyline, 2^20 points/iter, L=3, BCE + eikonal
  kitti-large  the same on an 8.4 km serpentine whose feature tables (> 256 MiB) do not fit the Infinity Cache
  ncd-incre    config 4: incremental mapping (shine_incre.py:86-195), N=4096, 50 iterations per frame, regulariser,
               fused Adam, importance sweep, octree growth on the device, one HIP graph per iteration
A "step" (batch workloads) is one pass of the hot path over one batch resident in HBM: sorted draw from the
node-ordered pool (also clears the dense grads, i.e. opt.zero_grad) -> fused query + decode + loss + backward
(shine_batch.py:123-209 minus the optimiser) [-> gradient exchange under data parallelism].  On one rank with batches of
<= 2^18 points (--draw-rider auto) the draw and the zero-fill of step i + 1 ride on the reduction launch of step i: a step is
two launches — the fused kernel and its reduction — over two alternating gradient buckets.  For ncd-incre a step is
one frame.  With N>1 ranks (torch.distributed.run, one process per GPU, RCCL): ONE global sorted draw (same seed
everywhere), rank r takes the r-th contiguous slice of P points (weak scaling), global normalisers come from the common
draw, and the grads are exchanged (--exchange).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches this script under `torch.distributed.run` with N
ranks on 127.0.0.1; fewer than N visible devices is an error, never a silent fall-back to one rank.

OUTPUT (rank 0, stdout).  The LAST line is the driver's record: ONE compact JSON object (< 4 kB; tests/test_bench_launch.py
holds the formatter to that) with the contract's fields + `roofline` + `cpu_baseline`.  The default invocation (no --workload,
one GPU) also runs abbreviated legs of BASELINE configs 3 (`kitti`), 4 (`ncd-incre`, Tier B and Tier A), the map beyond the
Infinity Cache (`kitti-large`) and rank 0's share of config 5 (`kitti-dp8-rank`); each prints its OWN compact line
(`{"leg": ...}`) BEFORE the record.  The full (un-cut) records go to --full-record-dir (default gpurun_out/bench_records/).
DESIGN.md §5 explains every field.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from shine_mapping_amd import benchlib  # noqa: E402
from shine_mapping_amd.benchlib import (WORKLOADS, gpu_iteration_n4096, kernel_roofline, run_batch,  # noqa: E402,F401
                                        run_dp_rank, run_incremental)

RECORD_LIMIT = 4096  # bytes of the final line (the 22 kB line of round 5 was not parsed by the driver)


def cpu_baseline(wl, seconds=12.0, n=4096, max_iters=400, regularize=False):
    """The oracle port of the reference's CPU path (same per-point dict lookups, the same torch CPU op sequence, the
    reference's Adam groups) on a bounded sample of the same workload: whole iterations — query + decode + loss +
    backward + Adam step (shine_batch.py:115-210, the reference's timing(s)/total) — of N=4096 points (the reference's
    own batch size, config/*/..._batch.yaml `bs`) drawn from the same pool / octree, for ~`seconds` of CPU work.
    regularize: the incremental configuration's iteration (shine_incre.py:152-158): + lambda_forget * cal_regularization on the
    octree's importance_weight / features_last_frame (an attached clone, model/feature_octree.py:160, as in a later frame)."""
    from oracle import shine_oracle as so

    cfg = wl.cfg
    ocfg = so.make_config(tree_level_world=cfg.tree_level_world, tree_level_feat=cfg.tree_level_feat,
                          leaf_vox_size=cfg.leaf_vox_size, sigma_sigmoid_m=cfg.sigma_sigmoid_m,
                          ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e,
                          loss_reduction=cfg.loss_reduction, lambda_forget=getattr(cfg, "lambda_forget", 0.0))
    oct_ = so.OracleOctree(ocfg)
    for lvl, tab in enumerate(wl.octree.nodes_lookup_tables):
        oct_.node_table[lvl] = tab
    oct_.hier_features = [p.detach().cpu().clone().requires_grad_(True) for p in wl.octree.hier_features]
    if regularize:
        oct_.importance_weight = [t.detach().cpu().clone() for t in wl.octree.importance_weight]
        oct_.features_last_frame = [p.clone() for p in oct_.hier_features]  # (:160: attached)
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
        so.train_step(oct_, mlp, c, l, w, ocfg, regularize=regularize)
        t1 = time.perf_counter()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return time.perf_counter() - t0, t1 - t0

    # torch's default (all host cores) is pathological for these small ops on a many-core host, so give the
    # CPU path its best thread count: the MEDIAN of three iterations per candidate (after a warm-up), then time with the
    # winner.  (One iteration per candidate let a single slow iteration pick the wrong count: the r02 figure moved 2x
    # between boxes.)
    calib = {}
    # (all cores only on small hosts: on the 256-core GPU box one iteration at 256 threads takes ~10 s — four of them were
    # 40 s of every run for a candidate that never wins)
    cands = {min(host_cores, 32), min(host_cores, 8), 1} | ({host_cores} if host_cores <= 64 else set())
    for threads in sorted(cands, reverse=True):
        torch.set_num_threads(threads)
        one_iteration()  # warm-up at this thread count
        calib[threads] = statistics.median(one_iteration()[0] for _ in range(3))
    best_threads = min(calib, key=calib.get)
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
        "thread_calibration_ms": {str(k): v * 1e3 for k, v in sorted(calib.items())},
        "sample_short": "%d whole iterations of N=%d (query+decode+loss%s+backward+Adam) from the same pool/octree, oracle port of "
                        "the reference CPU path, best of {32,8,1} threads" % (it, n, "+regulariser" if regularize else ""),
        "sample": "%d whole iterations (query+decode+loss%s+backward+Adam step, the reference's timing(s)/total) of N=%d "
                  "(reference batch size) from the same pool/octree; oracle/shine_oracle.py train_step + "
                  "torch.optim.Adam(betas=(0.9,0.99), eps=1e-15) on the reference's groups, torch %s CPU, thread count = "
                  "best median of 3 iterations over {32,8,1} (+ all cores on hosts of <= 64)" % (
                      it, "+lambda_forget*cal_regularization" if regularize else "", n, torch.__version__),
    }


# --------------------------------------------------------------------------------------------- the printed record


def _num(x, sig=5):
    """floats to `sig` significant digits (the line is for reading and parsing, the full record keeps everything)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return x


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_roofline(roof):
    if not isinstance(roof, dict):
        return None
    return _num({
        "bound": roof.get("bound"), "achieved": roof.get("achieved"), "peak": roof.get("peak"), "unit": roof.get("unit"),
        "frac": roof.get("frac"), "traffic": roof.get("traffic"), "kernel_ms": roof.get("kernel_ms"),
        "frac_8d": roof.get("frac_8d"), "bytes_per_point": _get(roof, "algorithmic", "bytes_per_point"),
        "hbm_frac": _get(roof, "hbm", "frac"), "compulsory_bytes": _get(roof, "hbm", "compulsory_bytes"),
        "table_bytes": _get(roof, "hbm", "table_bytes"), "in_infinity_cache": roof.get("map_in_infinity_cache"),
        "datapath_frac": _get(roof, "datapath", "frac"), "mfma_useful_frac": _get(roof, "datapath", "mfma_useful_frac"),
        "mfma_busy_pmc": _get(roof, "datapath", "mfma_busy_pmc"),
        "l2_atomics_frac": _get(roof, "l2_atomics", "frac_of_kernel_time"),
        "workgroups": _get(roof, "launch_geometry", "workgroups"), "pmc_used": _get(roof, "pmc", "used"),
    })


def compact_cpu_baseline(cb):
    if not isinstance(cb, dict):
        return None
    return _num({"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                 "host_cores": cb.get("host_cores"), "ms_per_iteration": cb.get("ms_per_iteration"),
                 "sample": cb.get("sample_short") or str(cb.get("sample"))[:200]})


def compact(rec, full_path=None):
    """the driver's record out of a leg's full record: the contract's fields + roofline + cpu_baseline, nothing nested deeper
    than two levels, < RECORD_LIMIT bytes"""
    if not isinstance(rec, dict) or "error" in rec:
        return rec
    cfg = rec.get("config", {})
    out = {k: rec.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {
        "workload": cfg.get("workload_short") or str(cfg.get("workload"))[:240],
        "points_per_iter_per_gpu": cfg.get("points_per_iter_per_gpu"), "levels": cfg.get("levels"),
        "pool_samples": cfg.get("pool_samples"), "corner_rows": cfg.get("corner_rows"),
        "feature_table_bytes": cfg.get("feature_table_bytes"), "parallelism": cfg.get("parallelism"),
        "launch": str(cfg.get("launch"))[:120],
        "window_ms": {k: _get(cfg, "window_ms", k) for k in ("median", "min", "max")} if cfg.get("window_ms") else None,
    }
    if rec.get("n_gpus", 1) and rec.get("n_gpus", 1) > 1:
        out["config"].update({"grad_exchange": str(cfg.get("grad_exchange"))[:240], "rank_ms_per_step": cfg.get("rank_ms_per_step"),
                              "world_size_reported": cfg.get("world_size_reported"),
                              "grad_exchange_overflow": cfg.get("grad_exchange_overflow")})
    out["roofline"] = compact_roofline(rec.get("roofline"))
    out["cpu_baseline"] = compact_cpu_baseline(rec.get("cpu_baseline"))
    lf = rec.get("like_for_like")
    if isinstance(lf, dict) and "speedup" in lf:
        out["like_for_like"] = {"n": lf.get("n"), "gpu_samples_per_s": _get(lf, "gpu", "samples_per_s") or lf.get("gpu_samples_per_s_in_loop"),
                                "cpu_samples_per_s": lf.get("cpu_samples_per_s"), "speedup": lf.get("speedup"),
                                "what": "same N and iteration definition (incl. Adam) on both sides"}
    it = rec.get("iteration_with_fused_adam")
    if isinstance(it, dict):
        out["iteration_with_fused_adam_ms"] = it.get("ms_per_iteration")
    for k in ("frames_per_s", "us_per_iteration", "final_loss"):
        if k in rec:
            out[k] = rec[k]
    ta = rec.get("tier_a")
    if isinstance(ta, dict):
        out["tier_a"] = ({"error": str(ta["error"])[:160]} if "error" in ta else
                         {k: ta.get(k) for k in ("frames_per_s", "ms_per_iteration", "vs_tier_b_frames_per_s", "autograd_nodes")})
    if full_path:
        out["full_record"] = full_path
    out = _num(out)
    line = json.dumps(out)
    if len(line) >= RECORD_LIMIT:  # never let a long string take the record down: cut the free-text fields, then drop extras
        out["config"]["workload"] = out["config"]["workload"][:120]
        for k in ("like_for_like", "tier_a", "iteration_with_fused_adam_ms"):
            out.pop(k, None)
        if out.get("cpu_baseline"):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:80]
    return out


def compact_dp_rank(rec):
    if not isinstance(rec, dict) or "error" in rec:
        return rec
    keep = ("points_per_rank", "n_global", "world", "ms_per_step_measured", "exchange_local_cost_ms", "message_rows_per_rank",
            "message_capacity_rows", "message_overflow", "dense_bucket_bytes", "reduced_vs_single_process_max_rel_err")
    out = {k: rec.get(k) for k in keep if k in rec}
    sm = _get(rec, "scale_model", "by_exchange")
    if isinstance(sm, dict):
        out["scale_model_MODELLED_ms_per_step"] = {k: v.get("ms_per_step") for k, v in sm.items()}
    return _num(out)


def write_full(directory, name, rec):
    """the un-cut record of a leg -> <directory>/<name>.json (best effort: the box may be read-only)"""
    try:
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, name + ".json")
        with open(path, "w") as f:
            json.dump(rec, f, indent=1, default=str)
        return os.path.relpath(path, ROOT)
    except Exception:
        return None


def emit(obj):
    """one JSON object = one stdout line, flushed (stderr chatter must not end up inside it)"""
    sys.stderr.flush()
    sys.stdout.write(json.dumps(obj, allow_nan=False, default=str) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="default 200 (ncd-incre: 12 frames)")
    ap.add_argument("--warmup", type=int, default=-1, help="default 20 (ncd-incre: 3 frames)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: maicity (BASELINE config 2) plus abbreviated kitti and ncd-incre legs under `configs`")
    ap.add_argument("--points", type=int, default=0, help="points per iteration per GPU (default: the workload's)")
    ap.add_argument("--levels", type=int, default=0, help="tree_level_feat (default: the workload's)")
    ap.add_argument("--frames", type=int, default=0, help="scans the synthetic map is built from")
    ap.add_argument("--iters", type=int, default=50, help="ncd-incre: iterations per frame (config iters)")
    ap.add_argument("--unroll", type=int, default=5,
                    help="ncd-incre: iterations per HIP graph.  The graph is built once and its nodes are re-bound every frame "
                         "(two parameter updates per iteration held, on the host's critical path) against ~16 us of idle GPU per "
                         "graph boundary: measured 351 / 390 / 378 frames/s at 2 / 5 / 10 (profiles/r04_ab_experiments.txt "
                         "blocks 3 and 9)")
    ap.add_argument("--sync-frames", action="store_true",
                    help="ncd-incre: one host synchronisation at the end of every frame (a frame's latency) instead of letting the "
                         "host run a frame ahead of the device (throughput)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of exactly --steps steps each; the line reports the median window")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tier-a", action="store_true", help="ncd-incre: skip the Tier A (unchanged driver's loop) measurement")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="default invocation: skip the abbreviated kitti / ncd-incre legs")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--draw-rider", default="auto", choices=["auto", "on", "off"],
                    help="one rank: the step's reduction launch also draws the next batch and clears the next step's bucket (a step = "
                         "two launches, two alternating gradient buckets) instead of a stand-alone draw launch in front of every "
                         "step.  auto: for batches of <= 2^18 points, where the draw is launch-bound (measured: -4 %% step time at "
                         "2^18, +1-3 %% at 2^20)")
    ap.add_argument("--preheat-ms", type=float, default=40.0,
                    help="untimed replays of the step before the warm-up steps, so that short runs see ramped-up clocks")
    ap.add_argument("--graph-steps", type=int, default=0,
                    help="steps captured per HIP graph (K steps = K // U replays + K %% U one-step replays); 0 = the largest "
                         "divisor of --steps up to 20")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="data parallel, gather exchange: fused steps per rank and step; with M > 1 the (asynchronous) "
                         "all-gather of one micro-batch overlaps the fused kernel of the next, and --exchange auto measures "
                         "that candidate too.  Default 1: the overlapped form is covered by the two-rank tests but has never "
                         "run on a multi-GPU node, and a collective that misbehaves under graph capture would take the "
                         "whole scaling measurement with it")
    ap.add_argument("--exchange", default="auto", choices=["auto", "gather", "dense", "touched"],
                    help="data-parallel gradient exchange: one flat all-reduce of the dense grads, or only the rows the "
                         "global batch touched (auto: touched when the dense bucket exceeds 64 MB)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the data-parallel code path even at world size 1")
    ap.add_argument("--full-record-dir", default=os.path.join(ROOT, "gpurun_out", "bench_records"),
                    help="where the un-cut record of every leg is written (the printed lines are cut to < 4 kB)")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher / rendezvous check only (no GPU work): used by tests/test_bench_launch.py")
    args = ap.parse_args()
    benchlib.maybe_spawn(args, sys.argv[1:], __file__)
    dist, world, rank, local_rank, backend = benchlib.init_ranks(args)
    if args.launch_check:
        return benchlib.launch_check(args, dist, world, rank, backend)

    default_run = args.workload is None
    if default_run:
        args.workload = "maicity"
    incre = args.workload == "ncd-incre"
    if args.steps <= 0:
        args.steps = 12 if incre else 200
    if args.warmup < 0:
        args.warmup = 3 if incre else 20
    if dist is None:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)
    cb = None if args.no_cpu_baseline else cpu_baseline
    full_dir = args.full_record_dir

    if incre:
        if world > 1:
            raise SystemExit("ncd-incre is a single-GPU workload (BASELINE.json config 4)")
        rec = run_incremental(args, dev, args.steps, args.warmup, cpu_baseline=cb)
        emit(compact(rec, write_full(full_dir, "ncd-incre", rec)))
        return

    rec = run_batch(args, args.workload, dist, world, rank, dev, args.steps, args.warmup, cpu_baseline=cb)
    if default_run and world == 1 and not args.no_extra_configs:
        # BASELINE configs 3, 4 and 5's rank share under the same driver clock: abbreviated legs (fewer steps, shorter CPU
        # samples), each printed as its own compact line BEFORE the record.  A failing leg is reported, never allowed to take
        # the headline line down.
        legs = (
            ("kitti", lambda: run_batch(args, "kitti", None, 1, 0, dev, 60, 10, cpu_baseline=cb, with_like_for_like=False,
                                        with_iteration=False, cpu_seconds=6.0)),
            ("ncd-incre", lambda: run_incremental(args, dev, 10, 3, cpu_baseline=cb, cpu_seconds=6.0)),
            # the regime real KITTI-00 lives in: feature tables beyond the Infinity Cache (the far build of the fused step).  No
            # CPU baseline here: its python dict of 10^7 nodes takes minutes (profiles/r04_bench_kitti-large.json.log has one)
            ("kitti-large", lambda: run_batch(args, "kitti-large", None, 1, 0, dev, 40, 10, cpu_baseline=None,
                                              with_like_for_like=False, with_iteration=False)),
            # config 5 as far as one GPU can measure it: rank 0's slice + the eight ranks' REAL messages
            ("kitti-dp8-rank", lambda: run_dp_rank(args, dev)),
        )
        for name, leg in legs:
            benchlib._release(dev)
            try:
                full = leg()
                line = compact_dp_rank(full) if name == "kitti-dp8-rank" else compact(full)
            except Exception as e:
                full = line = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            path = write_full(full_dir, name, full)
            emit({"leg": name, "full_record": path, **(line if isinstance(line, dict) else {"value": line})})
    if rank == 0:
        emit(compact(rec, write_full(full_dir, args.workload, rec)))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
