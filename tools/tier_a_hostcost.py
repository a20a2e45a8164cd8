#!/usr/bin/env python
"""Where Tier A's host time goes: the BCE loop of tools/tier_a_bench.py (shine_batch.py:115-210 verbatim on the drop-in's names,
C++ nodes, backward on the calling thread) with a host clock around every statement — no device synchronisation inside the loop, so
each figure is what the statement costs the ISSUING thread (at N = 4096 the device needs ~50 us per iteration, the host ~200: the
host is the bound) — then the same loop under cProfile (top functions by own time).  us per iteration, median of 5 x 200."""
import cProfile, os, pstats, statistics, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import autograd_ops, losses, optim, synth

os.environ["SHINE_TIER_A_EXT"] = "1"
torch.autograd.set_multithreading_enabled(False)
kind, n, lv = (sys.argv[1] if len(sys.argv) > 1 else "maicity"), 4096, 3
wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=lv)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
g = torch.Generator(device="cuda").manual_seed(1)
sigma = cfg.sigma_sigmoid
opt = optim.setup_optimizer(cfg, list(octree.parameters()), list(dec.parameters()))
NAMES = ["get_batch", "query_feature", "sdf", "mask+abs", "sdf_bce_loss", "0.+loss", "zero_grad", "backward", "opt.step"]
acc = [0.0] * len(NAMES)
pc = time.perf_counter


def loop(timed):
    t = [pc()]
    coord, sdf_label, weight = synth.draw_batch(wl.pool, n, g); t.append(pc())
    feature = octree.query_feature(coord); t.append(pc())
    sdf_pred = dec.sdf(feature); t.append(pc())
    surface_mask = weight > 0
    weight = torch.abs(weight); t.append(pc())
    l = losses.sdf_bce_loss(sdf_pred, sdf_label, sigma, weight, False, cfg.loss_reduction); t.append(pc())
    cur_loss = 0.
    cur_loss += l; t.append(pc())
    opt.zero_grad(set_to_none=True); t.append(pc())
    cur_loss.backward(); t.append(pc())
    opt.step(); t.append(pc())
    if timed:
        for i in range(len(NAMES)):
            acc[i] += t[i + 1] - t[i]


for _ in range(50):
    loop(False)
rows, totals = [], []
for rep in range(5):
    acc[:] = [0.0] * len(NAMES)
    torch.cuda.synchronize()
    t0 = pc()
    for _ in range(200):
        loop(True)
    torch.cuda.synchronize()
    totals.append((pc() - t0) / 200 * 1e6)
    rows.append([a / 200 * 1e6 for a in acc])
med = [statistics.median(r[i] for r in rows) for i in range(len(NAMES))]
print("%s L%d N=%d BCE, dropin C++ nodes, calling-thread backward: %.1f us per iteration (sum of statements %.1f)" %
      (kind, lv, n, statistics.median(totals), sum(med)))
for nm, m in zip(NAMES, med):
    print("  %-14s %6.1f us" % (nm, m))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    loop(False)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(22)
