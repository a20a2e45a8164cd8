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
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    n = 4096
    g = torch.Generator().manual_seed(123)
    pool_n = wl.pool.sdf_label.shape[0]
    done, t_used = 0, 0.0
    it = 0
    while t_used < seconds and it < 400:
        idx = torch.randint(0, pool_n, (n,), generator=g)
        c, l, w = wl.pool.coord[idx.to(wl.pool.coord.device)].cpu(), wl.pool.sdf_label[idx.to(wl.pool.coord.device)].cpu(), \
            wl.pool.weight[idx.to(wl.pool.coord.device)].cpu()
        t0 = time.perf_counter()
        so.train_step(oct_, mlp, c, l, w, ocfg)
        dt = time.perf_counter() - t0
        if it >= 2:  # two warm-up iterations
            done += n
            t_used += dt
        it += 1
    return {
        "value": done / max(t_used, 1e-9), "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": "%d iterations of N=4096 (reference batch size) from the same pool/octree; oracle/shine_oracle.py "
                  "train_step = query+decode+loss+backward, torch %s CPU" % (max(it - 2, 0), torch.__version__),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="maicity", choices=["maicity", "kitti"])
    ap.add_argument("--points", type=int, default=0, help="points per iteration per GPU (default: 2^18 maicity, 2^20 kitti)")
    ap.add_argument("--levels", type=int, default=0, help="tree_level_feat (default: 4 for maicity per BASELINE.json config 2, 3 for kitti)")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sort", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

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
    reducer = shine_dp.GradReducer(params, dist) if world > 1 else None

    ev_pairs = []

    def step(i, timed):
        c, l, w = batches[i % len(batches)]
        for p in params:
            p.grad.zero_()
        n_surf = None
        if opts.ekional_loss_on:
            n_surf = (w > 0).sum()
            if reducer is not None:
                reducer.all_reduce_scalar(n_surf)
        perm = None if args.no_sort else shine_dp.morton_order(octree, c)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss, pred, _ = fused_train_step(octree, decoder, c, l, w, opts, perm=perm, n_surf=n_surf)
        if timed:
            e1.record()
            ev_pairs.append((e0, e1))
        if reducer is not None:
            reducer.all_reduce_grads()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(len(ev_pairs), 1)

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
                "morton_sorted": not args.no_sort, "parallelism": "dp%d" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_point": bpp,
            },
            "final_loss": float(loss),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, wl)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
