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
pc = time.perf_counter


def incremental():
    """config 4's iteration (shine_incre.py:114-181 on the drop-in's names: + cal_regularization, sum reduction), frames >= 2"""
    from shine_mapping_amd import Decoder, FeatureOctree, incre_learning

    dev = "cuda"
    cfg = synth.make_config("ncd", device=dev, lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0, tree_level_feat=3)
    frames = list(synth.make_frames(cfg, frames=6, beams=64, azimuths=900, seed=42, device=dev))
    torch.manual_seed(0)
    octree, geo_mlp = FeatureOctree(cfg), Decoder(cfg)
    gen = torch.Generator(device=dev).manual_seed(1)
    names = ["get_batch", "query_feature", "sdf", "abs", "sdf_bce_loss(sum)", "0.+loss", "cal_regularization", "+= lambda * reg",
             "zero_grad", "backward", "opt.step"]
    acc, count = [0.0] * len(names), 0
    for fi, (coord, label, weight) in enumerate(frames):
        octree.update(coord[weight > 0], incremental_on=True)
        opt = optim.setup_optimizer(cfg, list(octree.parameters()), list(geo_mlp.parameters()))
        pool = type("P", (), {"coord": coord, "sdf_label": label, "weight": weight})()
        torch.cuda.synchronize()
        prof = None
        if os.environ.get("PROFILE") and fi == len(frames) - 1:  # cProfile over the last frame's iterations
            prof = cProfile.Profile()
            prof.enable()
        for it in range(50):
            t = [pc()]
            c, sdf_label, w = synth.draw_batch(pool, 4096, gen); t.append(pc())
            feature = octree.query_feature(c); t.append(pc())
            sdf_pred = geo_mlp.sdf(feature); t.append(pc())
            w = torch.abs(w); t.append(pc())
            sdf_loss = losses.sdf_bce_loss(sdf_pred, sdf_label, cfg.sigma_sigmoid, w, False, "sum"); t.append(pc())
            cur_loss = 0.
            cur_loss += sdf_loss; t.append(pc())
            reg_loss = octree.cal_regularization(); t.append(pc())
            cur_loss += cfg.lambda_forget * reg_loss; t.append(pc())
            opt.zero_grad(set_to_none=True); t.append(pc())
            cur_loss.backward(); t.append(pc())
            opt.step(); t.append(pc())
            if fi >= 2 and it >= 5:
                count += 1
                for i in range(len(names)):
                    acc[i] += t[i + 1] - t[i]
        torch.cuda.synchronize()
        if prof is not None:
            prof.disable()
            print("== cProfile of the last frame's 50 iterations (divide by 50), by own time")
            pstats.Stats(prof).sort_stats("tottime").print_stats(40)
        opt.zero_grad(set_to_none=True)
        data = type("D", (), {"coord_pool": coord, "sdf_label_pool": label})()
        incre_learning.cal_feature_importance(data, octree, geo_mlp, cfg.sigma_sigmoid, 4096, 2, "sum")
    print("ncd-incre (config 4) iteration, dropin C++ nodes, calling-thread backward, frames >= 2: %.1f us (sum of statements)" %
          (sum(acc) / count * 1e6))
    for nm, a in zip(names, acc):
        print("  %-20s %6.1f us" % (nm, a / count * 1e6))


def eikonal():
    """the eikonal loop (shine_batch.py:115-210 with ekional_loss_on) on the drop-in's names; `g[surface_mask]` (:183) makes the
    host wait for the device in every iteration"""
    n, lv = 4096, 3
    wl = synth.build_workload("kitti", frames=30, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
    g = torch.Generator(device="cuda").manual_seed(1)
    sigma = cfg.sigma_sigmoid
    opt = optim.setup_optimizer(cfg, list(octree.parameters()), list(dec.parameters()))
    names = ["get_batch", "requires_grad_", "query_feature", "sdf", "weight > 0", "get_gradient * sigma", "abs", "sdf_bce_loss",
             "g[surface_mask]  (sync)", "norm .. mean, += w_e *", "zero_grad", "backward", "opt.step"]
    acc, count = [0.0] * len(names), 0
    autograd_ops.FUSE_WITH_COORD_GRAD = True
    for it in range(350):
        t = [pc()]
        coord, sdf_label, weight = synth.draw_batch(wl.pool, n, g); t.append(pc())
        coord.requires_grad_(True); t.append(pc())
        feature = octree.query_feature(coord); t.append(pc())
        sdf_pred = dec.sdf(feature); t.append(pc())
        surface_mask = weight > 0; t.append(pc())
        gr = losses.get_gradient(coord, sdf_pred) * sigma; t.append(pc())
        weight = torch.abs(weight); t.append(pc())
        cur_loss = 0.
        cur_loss += losses.sdf_bce_loss(sdf_pred, sdf_label, sigma, weight, False, cfg.loss_reduction); t.append(pc())
        gs = gr[surface_mask]; t.append(pc())
        cur_loss += cfg.weight_e * ((gs.norm(2, dim=-1) - 1.0) ** 2).mean(); t.append(pc())
        opt.zero_grad(set_to_none=True); t.append(pc())
        cur_loss.backward(); t.append(pc())
        opt.step(); t.append(pc())
        if it >= 50:
            count += 1
            for i in range(len(names)):
                acc[i] += t[i + 1] - t[i]
    torch.cuda.synchronize()
    autograd_ops.FUSE_WITH_COORD_GRAD = False
    # the driver's own eikonal term (shine_batch.py:141-142, 182-185) on a LEAF tensor in place of get_gradient's result: torch only
    tt = [0.0, 0.0]
    for it in range(250):
        coord, sdf_label, weight = synth.draw_batch(wl.pool, n, g)
        leaf = torch.randn(n, 3, device="cuda", requires_grad=True)
        surface_mask = weight > 0
        torch.cuda.synchronize()
        t0 = pc()
        gr = leaf * sigma
        loss = cfg.weight_e * ((gr[surface_mask].norm(2, dim=-1) - 1.0) ** 2).mean()
        t1 = pc()
        loss.backward()
        t2 = pc()
        if it >= 50:
            tt[0] += t1 - t0
            tt[1] += t2 - t1
    print("the driver's eikonal term alone (torch ops on a leaf tensor, idle device): forward %.1f us, backward %.1f us" %
          (tt[0] / 200 * 1e6, tt[1] / 200 * 1e6))
    print("kitti L3 N=4096 BCE + eikonal, dropin C++ nodes, calling-thread backward: %.1f us (sum of statements)" % (sum(acc) / count * 1e6))
    for nm, a in zip(names, acc):
        print("  %-26s %6.1f us" % (nm, a / count * 1e6))


if len(sys.argv) > 1 and sys.argv[1] == "incre":
    incremental()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "eik":
    eikonal()
    sys.exit(0)
kind, n, lv = (sys.argv[1] if len(sys.argv) > 1 else "maicity"), 4096, 3
wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=lv)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
g = torch.Generator(device="cuda").manual_seed(1)
sigma = cfg.sigma_sigmoid
opt = optim.setup_optimizer(cfg, list(octree.parameters()), list(dec.parameters()))
NAMES = ["get_batch", "query_feature", "sdf", "mask+abs", "sdf_bce_loss", "0.+loss", "zero_grad", "backward", "opt.step"]
acc = [0.0] * len(NAMES)


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
