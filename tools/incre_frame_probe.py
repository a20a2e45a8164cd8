#!/usr/bin/env python
"""Where a slow frame of the incremental loop spends its time: the iteration phase of bench.run_incremental split into
{optimiser state + flags + object, bind (set_step / set_finish / commit), launch + wait}, per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, synth
from shine_mapping_amd.loop import GraphedIteration
from shine_mapping_amd.optim import setup_optimizer
from shine_mapping_amd.sampler import SortedPool

dev = torch.device("cuda")
cfg = synth.make_config("ncd", device=dev, lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0, tree_level_feat=3)
frames = list(synth.make_frames(cfg, frames=15, beams=64, azimuths=900, seed=42, device=dev))
torch.manual_seed(0)
octree, dec = FeatureOctree(cfg), Decoder(cfg)
opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum")
native = os.environ.get("NATIVE", "1") == "1"
for fi, (coord, label, weight) in enumerate(frames):
    octree.update(coord[weight > 0], incremental_on=True)
    octree._require_tables(with_ranks=True)
    opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
    pool = SortedPool(octree, coord, label, weight, seed=fi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step = GraphedIteration(octree, dec, pool, opt, opts, 4096, lambda_forget=cfg.lambda_forget, unroll=10, eager_first=fi == 0,
                            native=native)
    tc = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if t1 - t0 > 5e-3:
        print("   (construct: host %.3f ms, then sync %.3f ms)" % ((tc - t0) * 1e3, (t1 - tc) * 1e3))
    if native:
        step._bound(10)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    step.run(49 if step.ran_eager else 50)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    if os.environ.get("SWEEP", "1") == "1":
        from shine_mapping_amd.incre_learning import cal_feature_importance
        data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
        cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, 4096, 2, "sum", pool=pool)
        torch.cuda.synchronize()
    t5 = time.perf_counter()
    print("frame %2d rows %s: construct %.3f ms, bind %.3f, launch calls %.3f, wait %.3f, sweep %.3f" % (
        fi, [int(p.shape[0]) for p in octree.hier_features], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
        (t5 - t4) * 1e3), flush=True)
