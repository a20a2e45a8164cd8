#!/usr/bin/env python
"""bench.py — trained SDF samples/s (fwd+bwd) of the fused SHINE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload maicity|kitti] [--points P] [--levels L]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
zero the dense grads (what opt.zero_grad + autograd's fresh grads amount to), [Morton-sort the batch],
then the fused query + decode + loss + backward (shine_batch.py:123-209 minus the optimiser).  With N>1
ranks (torch.distributed.run, one process per GPU, RCCL) every rank processes its own batch of P points
against replicated tables and the dense grads are all-reduced (weak scaling).

One JSON line on rank 0; see README/DESIGN.md for the roofline and cpu_baseline objects.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md chip table (spec; 6290 measured copy)


def algorithmic_bytes_per_point(levels: int, feat: int = 8) -> int:
    """24 + L*(40 + 2*8*F*4): batch in/out, one node record and 8 corner rows read + written per level."""
    return 24 + levels * (40 + 2 * 8 * feat * 4)


def measured_traffic(levels, points, eikonal):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected separately, gfx950 corrections applied — profiles/r01_pmc_traffic_*.json says how).  bench.py cannot run
    the profiler on itself, so the figure is only reported for the configuration it was measured on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic_v1_bce_2p18_L4.json")
    if eikonal or levels != 4 or points != (1 << 18) or not os.path.isfile(path):
        return None
    try:
        return float(json.load(open(path))["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(args, wl, seconds=12.0):
    """The oracle port of the reference's CPU path (same dict-lookup structure, torch CPU ops, Adam excluded
    like the GPU figure) on a bounded sample of the same workload: batches of 4096 points (the reference's own
    batch size, config/maicity/maicity_batch.yaml:54) drawn from the same pool, for ~`seconds` of CPU work."""
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
    host_cores = os.cpu_count() or 1
    n = 4096
    g = torch.Generator().manual_seed(123)
    pool_n = wl.pool.sdf_label.shape[0]
    pdev = wl.pool.coord.device

    def one_iteration():
        idx = torch.randint(0, pool_n, (n,), generator=g).to(pdev)
        c, l, w = wl.pool.coord[idx].cpu(), wl.pool.sdf_label[idx].cpu(), wl.pool.weight[idx].cpu()
        t0 = time.perf_counter()
        so.train_step(oct_, mlp, c, l, w, ocfg)
        return time.perf_counter() - t0

    # torch's default (all host cores) is pathological for these small ops on a many-core host, so give the
    # CPU path its best thread count: calibrate on one iteration each, then time with the winner.
    best_t, best_threads = None, 1
    for threads in sorted({host_cores, min(host_cores, 32), min(host_cores, 8), 1}, reverse=True):
        torch.set_num_threads(threads)
        one_iteration()  # warm-up at this thread count
        dt = one_iteration()
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, threads
    torch.set_num_threads(best_threads)
    done, t_used, it = 0, 0.0, 0
    while t_used < seconds and it < 400:
        t_used += one_iteration()
        done += n
        it += 1
    return {
        "value": done / max(t_used, 1e-9), "unit": "samples/s", "cores": best_threads, "kind": "port",
        "host_cores": host_cores,
        "sample": "%d iterations of N=4096 (reference batch size, config/maicity/maicity_batch.yaml:54) from the same "
                  "pool/octree; oracle/shine_oracle.py train_step = query+decode+loss+backward (no optimiser), torch %s "
                  "CPU, thread count calibrated over {all,32,8,1}" % (it, torch.__version__),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="maicity", choices=["maicity", "kitti"])
    ap.add_argument("--points", type=int, default=0, help="points per iteration per GPU (default: 2^18 maicity, 2^20 kitti)")
    ap.add_argument("--levels", type=int, default=0, help="tree_level_feat (default: 4 for maicity per BASELINE.json config 2, 3 for kitti)")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--order", default="plan", choices=["plan", "radix"],
                    help="plan: counting sort by octree node + slot hand-off (shine_plan_batch); radix: Morton radix sort")
    ap.add_argument("--sampler", default="pool", choices=["pool", "batch"],
                    help="pool: the step draws its batch as sorted indices from the node-ordered pool (sampler + order "
                         "fusion, f-3); batch: the step is handed pre-drawn unsorted batches and orders them itself")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the data-parallel code path even at world size 1")
    ap.add_argument("--no-overlap", action="store_true",
                    help="do not overlap the plan of batch i+1 with the fused step of batch i (second stream)")
    args = ap.parse_args()

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

    from shine_mapping_amd import StepOptions, fused_train_step, synth
    from shine_mapping_amd import dp as shine_dp

    levels = args.levels or (4 if args.workload == "maicity" else 3)
    points = args.points or ((1 << 18) if args.workload == "maicity" else (1 << 20))
    wl = synth.build_workload(args.workload, frames=args.frames, device=dev, seed=42, tree_level_feat=levels)
    cfg = wl.cfg
    octree, decoder, pool = wl.octree, wl.decoder, wl.pool
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction,
                       ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, n_global=points * world)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    total = args.steps + args.warmup
    batches = [synth.draw_batch(pool, points, gen) for _ in range(min(total, 8))]
    params = list(octree.hier_features) + decoder.fused_params()
    for p in params:
        p.grad = torch.zeros_like(p)
    reducer = shine_dp.GradReducer(params, dist)  # flat grad bucket: one fill per step, one all-reduce under DP

    def order(c, zero=None):
        if args.no_sort or args.order != "plan":
            if zero is not None:
                reducer.zero_grads()
            return (None if args.no_sort else shine_dp.morton_order(octree, c)), None
        return shine_dp.plan_batch(octree, c, zero=zero)  # the plan pass also clears the gradient bucket

    spool = None
    if args.sampler == "pool" and not args.no_sort:
        from shine_mapping_amd.sampler import SortedPool

        octree._require_tables(with_ranks=True)
        spool = SortedPool(octree, pool.coord, pool.sdf_label, pool.weight, seed=1000 + rank)  # once per frame

    def pool_step_body(i):
        """draw a sorted batch from the node-ordered pool (+ clear grads in the same pass) -> fused step (-> all-reduce)"""
        idx = spool.draw(points, zero=reducer.flat)
        n_surf = None
        if opts.ekional_loss_on:
            n_surf = (spool.weight[idx.long()] > 0).sum()
            reducer.all_reduce_scalar(n_surf)
        loss, pred, _ = fused_train_step(octree, decoder, None, None, None, opts, n_surf=n_surf, pool=spool, idx=idx)
        if use_dist:
            reducer.all_reduce_grads()
        return loss

    def step_body(i):
        """zero grads -> Morton order -> fused query+decode+loss+backward (-> all-reduce)"""
        c, l, w = batches[i % len(batches)]
        n_surf = None
        if opts.ekional_loss_on:
            n_surf = (w > 0).sum()
            reducer.all_reduce_scalar(n_surf)
        perm, slots = order(c, zero=reducer.flat)
        loss, pred, _ = fused_train_step(octree, decoder, c, l, w, opts, perm=perm, n_surf=n_surf, slots=slots)
        if use_dist:
            reducer.all_reduce_grads()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The loop body has no host sync and no allocation outside torch's allocator, so it is captured into HIP graphs
    # and replayed (launch-bound inner loops belong in hipGraphs); the collective stays outside the graph.
    #
    # Pipelined form (default, single GPU): the plan of batch i+1 (counting sort by node + slot lookup + clearing the
    # NEXT gradient bucket) depends only on that batch and the static tables, so it runs on a second stream while the
    # fused step of batch i — which waits on memory ~45 % of the time — owns the first.  Plan outputs and gradient
    # buckets are double-buffered; every iteration still does all of its own work inside the timed region (the plan
    # of the first timed batch is produced by the last warm-up iteration, the last timed iteration plans one ahead).
    launch = "eager"
    graphs, graph_loss = [], []
    pipelined = (spool is None and not args.no_graph and not args.no_overlap and not use_dist and not args.no_sort
                 and args.order == "plan" and len(batches) % 2 == 0)
    if pipelined:
        try:
            side = torch.cuda.Stream()
            nb = len(batches)
            total_p = sum(p.numel() for p in params)
            flats = [torch.zeros((total_p + 3) // 4 * 4, dtype=torch.float32, device=dev) for _ in range(2)]
            views = []
            for f in flats:
                vs, off = [], 0
                for p in params:
                    vs.append(f[off: off + p.numel()].view_as(p))
                    off += p.numel()
                views.append(vs)
            plans = [(torch.empty(points, dtype=torch.int32, device=dev),
                      torch.empty((points, levels), dtype=torch.int32, device=dev)) for _ in range(2)]

            def use_bucket(k):
                for p, v in zip(params, views[k]):
                    p.grad = v

            def pipelined_body(i):
                cur = torch.cuda.current_stream()
                c, l, w = batches[i % nb]
                cn = batches[(i + 1) % nb][0]
                side.wait_stream(cur)
                with torch.cuda.stream(side):  # plan of the NEXT batch, clears the NEXT gradient bucket
                    shine_dp.plan_batch(octree, cn, zero=flats[(i + 1) % 2], out=plans[(i + 1) % 2])
                use_bucket(i % 2)
                n_surf = (w > 0).sum() if opts.ekional_loss_on else None
                loss, _, _ = fused_train_step(octree, decoder, c, l, w, opts, perm=plans[i % 2][0], n_surf=n_surf,
                                              slots=plans[i % 2][1])
                cur.wait_stream(side)
                return loss

            shine_dp.plan_batch(octree, batches[0][0], zero=flats[0], out=plans[0])  # pipeline prologue
            for i in range(nb):
                pipelined_body(i)  # warm caches / allocate workspaces outside capture
            torch.cuda.synchronize()
            for i in range(nb):
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    graph_loss.append(pipelined_body(i))
                graphs.append(g_)
            launch = "hipgraph, plan(i+1) || fused(i) on two streams"
        except Exception as e:
            print("pipelined capture failed (%s); falling back" % e, file=sys.stderr)
            graphs, graph_loss, launch, pipelined = [], [], "eager", False
            torch.cuda.synchronize()
            for p in params:
                p.grad = torch.zeros_like(p)
            reducer = shine_dp.GradReducer(params, dist)
    if spool is not None:
        step_body = pool_step_body  # noqa: F811  (the pool-mode body replaces the batch-mode one everywhere below)
    if spool is not None and not args.no_graph and not use_dist:
        # a replayed graph would replay the same random stream id; graphs are captured for a ring of stream ids instead
        # (SortedPool.draws advances at capture time), i.e. the timed loop cycles through `len(batches)` distinct draws.
        pass
    if not pipelined and not args.no_graph and not use_dist:
        try:
            for i in range(len(batches)):
                step_body(i)  # warm caches / allocate workspaces outside capture
            torch.cuda.synchronize()
            for i in range(len(batches)):
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    graph_loss.append(step_body(i))
                graphs.append(g_)
            launch = "hipgraph"
        except Exception as e:  # capture not possible on this stack: measure eagerly and say so
            print("graph capture failed (%s); falling back to eager launches" % e, file=sys.stderr)
            graphs, graph_loss, launch = [], [], "eager"
            torch.cuda.synchronize()

    def step(i):
        if graphs:
            graphs[i % len(graphs)].replay()
            return graph_loss[i % len(graphs)]
        return step_body(i)

    # (pipelined mode: graph i consumes the plan graph i-1 produced, so iterations must run in order starting at 0)
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

    # dominant kernel: HIP events on the launch stream around R back-to-back launches of the fused kernel ALONE
    # (kernel_variant bit 0x2000 skips the 23-workgroup partial-sum reduction launch, so the bracket holds exactly
    # what rocprofv3 reports for shine::k_step_v1), averaged per launch.
    R = 10
    import copy
    kopts = copy.copy(opts)
    kopts.kernel_variant = 0x2000
    c0, l0, w0 = batches[0]
    if spool is not None:
        idx0 = spool.draw(points)
        ns0 = (spool.weight[idx0.long()] > 0).sum() if opts.ekional_loss_on else None

        def fused_only():
            for _ in range(R):
                fused_train_step(octree, decoder, None, None, None, kopts, n_surf=ns0, pool=spool, idx=idx0)
    else:
        perm0, slots0 = order(c0)
        ns0 = (w0 > 0).sum() if opts.ekional_loss_on else None

        def fused_only():
            for _ in range(R):
                fused_train_step(octree, decoder, c0, l0, w0, kopts, perm=perm0, n_surf=ns0, slots=slots0)

    fused_only()
    torch.cuda.synchronize()
    kg = None
    if launch == "hipgraph":
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

    # the reference's whole iteration (timing(s)/total, shine_batch.py:225): step + optimiser.  Fused Adam clears the
    # grads in the same pass, so the plan no longer has to.  Reported next to `value`, never instead of it.
    iter_ms = None
    if not use_dist:
        from shine_mapping_amd.optim import setup_optimizer

        cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
        for p in params:
            p.grad = torch.zeros_like(p)
        adam = setup_optimizer(cfg, list(octree.parameters()), decoder.fused_params())

        def iteration(i):
            if spool is not None:
                ix = spool.draw(points)
                ns = (spool.weight[ix.long()] > 0).sum() if opts.ekional_loss_on else None
                fused_train_step(octree, decoder, None, None, None, opts, n_surf=ns, pool=spool, idx=ix)
            else:
                c, l, w = batches[i % len(batches)]
                ns = (w > 0).sum() if opts.ekional_loss_on else None
                pm, sl = order(c)
                fused_train_step(octree, decoder, c, l, w, opts, perm=pm, n_surf=ns, slots=sl)
            adam.step(zero_grad=True)

        for i in range(3):
            iteration(i)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for i in range(args.steps):
            iteration(i)
        torch.cuda.synchronize()
        iter_ms = (time.perf_counter() - ti) / args.steps * 1e3

    if rank == 0:
        bpp = algorithmic_bytes_per_point(levels)
        achieved = points * bpp / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "trained SDF samples/sec (fwd+bwd)", "value": points * world * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s-like street canyon, batch mode, %d points/iter/GPU, %d-level octree (levels %d..%d), "
                            "F=8, decoder 8-32-32-1, %s" % (args.workload, points, levels,
                                                            cfg.tree_level_world - levels + 1, cfg.tree_level_world,
                                                            "BCE+eikonal" if cfg.ekional_loss_on else "BCE"),
                "points_per_iter_per_gpu": points, "levels": levels, "frames": args.frames,
                "pool_samples": int(pool.sdf_label.shape[0]),
                "corner_rows": [int(p.shape[0]) for p in octree.hier_features],
                "batch_order": "none" if args.no_sort else ("sorted draw from the node-ordered pool (f-3)"
                                                            if spool is not None else args.order),
                "parallelism": "dp%d" % world,
                "launch": launch,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(levels, points, bool(cfg.ekional_loss_on)), "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_point": bpp,
                # also against the measured copy bandwidth, and the compulsory traffic of one launch
                # (every feature row read once + its gradient row written once + the batch in/out)
                "frac_of_measured_copy_6290GBs": achieved / 6290.0,
                "compulsory_bytes": int(sum(int(p.shape[0]) for p in octree.hier_features) * 32 * 2 + 24 * points),
            },
            "final_loss": float(loss),
            "iteration_with_fused_adam": None if iter_ms is None else {
                "ms_per_iteration": iter_ms, "samples_per_s": points / (iter_ms * 1e-3), "launch": "eager",
                "what": "sorted draw (or plan) + fused step + fused dense Adam (also clears grads); reference timing(s)/total"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, wl)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
