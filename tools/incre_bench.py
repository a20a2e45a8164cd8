#!/usr/bin/env python
"""Incremental mapping with the forgetting regulariser (BASELINE config 4, shine_incre.py:86-195) on the fused path:

    per frame:  octree.update (device growth)  ->  optimiser re-creation  ->  pool plan  ->
                `iters` x { sorted draw of N, fused step (sum reduction, touched rows), regulariser, fused Adam }  ->
                importance sweep (cal_feature_importance, utils/incre_learning.py:8-40)

NCD-like synthetic quad (40 m, circular trajectory), N=4096, 50 iterations per frame, lambda_forget 1e4
(config/ncd/ncd_incre_reg.yaml).  Prints the per-frame time split and frames/s.  (The CPU-oracle baseline for the same
frames is timed by tests/cpu_baselines.py: only tests/ may touch oracle/.)
"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, fused_train_step, synth
from shine_mapping_amd.incre_learning import cal_feature_importance
from shine_mapping_amd.ops import fused_regularization, touched_flags
from shine_mapping_amd.optim import setup_optimizer
from shine_mapping_amd.sampler import SortedPool

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--bs", type=int, default=4096)
ap.add_argument("--freeze-after", type=int, default=20)
ap.add_argument("--graph", action="store_true", help="replay one captured HIP graph per iteration (loop.GraphedIteration)")
args = ap.parse_args()

cfg = synth.make_config("ncd", device="cuda", lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0)
frames = list(synth.make_frames(cfg, frames=args.frames, beams=64, azimuths=900, seed=42, device="cuda"))
print("frames=%d samples/frame~%d iters/frame=%d bs=%d" % (len(frames), int(np.mean([f[0].shape[0] for f in frames])),
                                                          args.iters, args.bs))


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def run():
    torch.manual_seed(0)
    octree, dec = FeatureOctree(cfg), Decoder(cfg).cuda()
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum")
    split = np.zeros((len(frames), 5))
    for fi, (coord, label, weight) in enumerate(frames):
        t0 = sync()
        octree.update(coord[weight > 0], incremental_on=True)
        octree._require_tables(with_ranks=True)
        t1 = sync()
        if fi == args.freeze_after:  # shine_incre.py:100-104: decoder frozen after the first frames
            for p in dec.parameters():
                p.requires_grad_(False)
            opts.decoder_grad_on = False
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
        pool = SortedPool(octree, coord, label, weight, seed=fi)
        touched = touched_flags(octree)
        t2 = sync()
        if args.graph:
            from shine_mapping_amd.loop import GraphedIteration
            step = GraphedIteration(octree, dec, pool, opt, opts, args.bs, lambda_forget=cfg.lambda_forget)  # = iteration 1
            for it in range(args.iters - 1):
                loss = step()
        for it in range(0 if args.graph else args.iters):
            idx = pool.draw(args.bs)
            loss, pred, _ = fused_train_step(octree, dec, None, None, None, opts, pool=pool, idx=idx, touched=touched)
            reg = fused_regularization(octree, cfg.lambda_forget, touched)
            opt.step(zero_grad=True)
        t3 = sync()
        data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
        cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, args.bs, 2, "sum")
        t4 = sync()
        split[fi] = (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)
    return octree, split * 1e3, float(loss)


run()
octree, split, loss = run()
med = np.median(split[1:], axis=0)
print("per frame (median, ms): update+ranks %.2f | optimiser+pool plan %.2f | %d iterations %.2f (%.1f us/iter) | "
      "importance sweep %.2f | total %.2f  -> %.1f frames/s, %.2f M trained samples/s in the loop" % (
          med[0], med[1], args.iters, med[2], med[2] / args.iters * 1e3, med[3], med[4], 1e3 / med[4],
          args.iters * args.bs / med[2] / 1e3))
print("rows %s final loss %.4f" % ([int(p.shape[0]) for p in octree.hier_features], loss))
