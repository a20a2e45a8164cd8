#!/usr/bin/env python
"""Steady-state iteration rate at the reference's own batch size (4096, config/maicity/maicity_batch.yaml:54) on a fixed
map (batch mode): eager launches vs one captured HIP graph per iteration (loop.GraphedIteration)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.loop import GraphedIteration
from shine_mapping_amd.optim import setup_optimizer
from shine_mapping_amd.sampler import SortedPool

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = 2000
wl = synth.build_workload("maicity", frames=60, device="cuda", seed=42, tree_level_feat=3)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
opts = StepOptions(sigma=cfg.sigma_sigmoid)
octree._require_tables(with_ranks=True)
pool = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())


def eager():
    idx = pool.draw(n)
    loss, _, _ = fused_train_step(octree, dec, None, None, None, opts, pool=pool, idx=idx)
    opt.step(zero_grad=True)
    return loss


for _ in range(50):
    eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    loss = eager()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / iters
t0 = time.perf_counter()
step = GraphedIteration(octree, dec, pool, opt, opts, n)
torch.cuda.synchronize()
tc = time.perf_counter() - t0
for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    loss = step()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / iters
print("N=%d rows=%s: eager %.1f us/iter (%.1f M samples/s) | graph replay %.1f us/iter (%.1f M samples/s), capture %.1f ms | loss %.4f" % (
    n, [int(p.shape[0]) for p in octree.hier_features], te * 1e6, n / te / 1e6, tg * 1e6, n / tg / 1e6, tc * 1e3, float(loss)))
