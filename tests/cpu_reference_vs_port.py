#!/usr/bin/env python
"""The CPU baseline bench.py reports is the oracle PORT (`cpu_baseline.kind: "port"`): the reference itself cannot travel to the
GPU box.  This script (test infrastructure: it runs the oracle and the REAL reference from /root/reference, authoring container
only) times both on the SAME cores, the same scene, the same batches and the same iteration definition — query_feature -> sdf ->
sdf_bce_loss -> backward -> Adam step at N = 4096 (shine_batch.py:115-210) — so that the port's figure can be read as the
reference's: same torch op sequence (the oracle is bit-identical, oracle/make_golden.py), same per-point dict lookups.

    python tests/cpu_reference_vs_port.py [threads] > profiles/r06_cpu_reference_vs_port.txt
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as mg  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle import shine_oracle as so  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(threads)
R = ref_import.install()
spec = dict(tree_level_world=12, tree_level_feat=3, leaf_vox_size=0.2, sigma_sigmoid_m=0.05, surface_sample_range_m=0.15,
            surface_sample_n=3, free_sample_n=3, free_sample_end_dist_m=0.8)
torch.manual_seed(1234)
cfg = mg.ref_config(R, spec)
ocfg = so.make_config(**spec)
sampler = R.dataSampler(cfg)
octree, mlp = R.FeatureOctree(cfg), R.Decoder(cfg)
oct2, mlp2 = so.OracleOctree(ocfg), so.OracleDecoder(ocfg)
mlp2.load_state_dict(mlp.state_dict())
coords, labels, weights = [], [], []
for f in range(6):  # six small scans of the fixture generator side by side: ~50 k samples
    pts, origin = mg.tiny_scan(100 + f, 1500, shift=2.5 * f)
    c, l, _, _, w, _, _ = sampler.sample(pts * cfg.scale, origin * cfg.scale, None, None)
    rs = torch.random.get_rng_state()
    octree.update(c[w > 0, :], False)
    torch.random.set_rng_state(rs)
    oct2.update(c[w > 0, :], False)
    coords.append(c), labels.append(l), weights.append(w)
coord, label, weight = torch.cat(coords), torch.cat(labels), torch.cat(weights)
sig = cfg.logistic_gaussian_ratio * cfg.sigma_sigmoid_m * cfg.scale
n, iters = 4096, 40
g = torch.Generator().manual_seed(7)
batches = [torch.randint(0, coord.shape[0], (n,), generator=g) for _ in range(iters + 3)]

cfg.opt_adam, cfg.lr = True, 0.01
opt_ref = R.setup_optimizer(cfg, list(octree.parameters()), list(mlp.parameters()), None, None)
opt_port = so.adam_param_groups(oct2, mlp2, 0.01)


def ref_iteration(idx):
    c, l, w = coord[idx], label[idx], weight[idx]
    feat = octree.query_feature(c)
    pred = mlp.sdf(feat)
    loss = R.sdf_bce_loss(pred, l, sig, torch.abs(w), cfg.loss_weight_on, cfg.loss_reduction)
    opt_ref.zero_grad(set_to_none=True)
    loss.backward()
    opt_ref.step()
    return float(loss.detach())


def port_iteration(idx):
    c, l, w = coord[idx], label[idx], weight[idx]
    out = so.train_step(oct2, mlp2, c, l, w, ocfg)
    opt_port.step()
    opt_port.zero_grad(set_to_none=True)
    return float(out["loss"])


def timed(fn):
    for idx in batches[:3]:
        fn(idx)
    t0 = time.perf_counter()
    last = None
    for idx in batches[3:]:
        last = fn(idx)
    dt = time.perf_counter() - t0
    return n * iters / dt, dt / iters * 1e3, last


rows = [int(p.shape[0]) for p in octree.hier_features]
print("scene: %d samples, corner rows %s, N = %d, %d timed iterations, torch %s, %d threads, %d host cores" % (
    coord.shape[0], rows, n, iters, torch.__version__, threads, os.cpu_count()))
for rep in range(2):
    a = timed(ref_iteration)
    b = timed(port_iteration)
    print("pass %d: REFERENCE modules (/root/reference + kaolin shim) %8.0f samples/s (%.2f ms / iteration, loss %.6f) | "
          "oracle PORT %8.0f samples/s (%.2f ms / iteration, loss %.6f) | port / reference = %.3f" % (
              rep, a[0], a[1], a[2], b[0], b[1], b[2], b[0] / a[0]))
