#!/usr/bin/env python
"""Tier A (strict drop-in) iteration time: the reference's loop body (shine_batch.py:115-210) VERBATIM on this package's
classes — query_feature -> sdf -> [get_gradient] -> sdf_bce_loss [+ eikonal] -> zero_grad / backward / opt.step —

  "utils unpatched"  what round 3 gave the unchanged driver: model.* replaced, utils.* the reference's own (torch composites
                     for the loss and get_gradient — so the eikonal loop runs the split, twice-differentiable nodes — and
                     torch.optim.Adam);
  "dropin"           what `import shine_mapping_amd.dropin` gives it now: utils.tools.setup_optimizer / get_gradient and
                     utils.loss.sdf_bce_loss re-bound to the fused forms (one-launch Adam, one-launch loss, the eikonal loop
                     on the fused node) — with the autograd nodes in Python (round 4) and in the C++ extension
                     (csrc/shine_torch_ext.cpp, round 5: SHINE_TIER_A_EXT);
  "tier B"           the fused step on the same unordered batch + the fused Adam, for scale.
Same process, same box, interleaved repetitions; ms per iteration = median of 5 x 30 iterations."""
import os, statistics, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, autograd_ops, fused_train_step, losses, optim, synth


def ref_get_gradient(inputs, outputs):  # utils/tools.py:175-185
    d = torch.ones_like(outputs, requires_grad=False)
    return torch.autograd.grad(outputs=outputs, inputs=inputs, grad_outputs=d, create_graph=True, retain_graph=True,
                               only_inputs=True)[0]


def tier_a_incremental(dev, frames, bs=4096, iters=50, warmup=2, levels=3, ext="1", single_thread=True):
    """BASELINE config 4 through the UNCHANGED driver's names: the loop body of shine_incre.py:100-195 verbatim
    on what `import shine_mapping_amd.dropin` binds them to — per frame  octree.update(incremental_on=True)  ->  setup_optimizer
    (a new Adam, :107-109)  ->  `iters` x {get_batch, query_feature, sdf, sdf_bce_loss(sum), lambda_forget * cal_regularization(),
    zero_grad / backward / step}  ->  cal_feature_importance — every launch issued eagerly by Python, one frame after the other
    (the driver synchronises for its timers between the phases, T0..T3).  `frames`: list of (coord, label, weight) per scan.
    -> dict(frames_per_s, ms_per_frame, ms_per_iteration, split)"""
    from shine_mapping_amd import Decoder, FeatureOctree, incre_learning

    os.environ["SHINE_TIER_A_EXT"] = ext
    mt_before = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(not single_thread)  # (dropin: backward on the calling thread)
    cfg = synth.make_config("ncd", device=dev, lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0, tree_level_feat=levels)
    torch.manual_seed(0)
    octree, geo_mlp = FeatureOctree(cfg), Decoder(cfg)
    sigma_sigmoid = cfg.sigma_sigmoid
    gen = torch.Generator(device=dev).manual_seed(1)
    t_frames, t_iter, t_update, t_sweep = [], [], [], []
    for fi, (coord, label, weight) in enumerate(frames):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        octree.update(coord[weight > 0], incremental_on=True)                       # dataset.process_frame, lidar_dataset.py:215
        octree_feat = list(octree.parameters())
        opt = optim.setup_optimizer(cfg, octree_feat, list(geo_mlp.parameters()))   # shine_incre.py:107-109
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pool = type("P", (), {"coord": coord, "sdf_label": label, "weight": weight})()
        for _ in range(iters):                                                      # shine_incre.py:114-181
            c, sdf_label, w = synth.draw_batch(pool, bs, gen)
            feature = octree.query_feature(c)
            sdf_pred = geo_mlp.sdf(feature)
            cur_loss = 0.
            w = torch.abs(w)
            sdf_loss = losses.sdf_bce_loss(sdf_pred, sdf_label, sigma_sigmoid, w, False, "sum")
            cur_loss += sdf_loss
            reg_loss = octree.cal_regularization()
            cur_loss += cfg.lambda_forget * reg_loss
            opt.zero_grad(set_to_none=True)
            cur_loss.backward()
            opt.step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.zero_grad(set_to_none=True)                                             # shine_incre.py:185-188
        data = type("D", (), {"coord_pool": coord, "sdf_label_pool": label})()
        incre_learning.cal_feature_importance(data, octree, geo_mlp, sigma_sigmoid, bs, 2, "sum")
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if fi >= warmup:
            t_frames.append(t3 - t0), t_update.append(t1 - t0), t_iter.append((t2 - t1) / iters), t_sweep.append(t3 - t2)
    torch.autograd.set_multithreading_enabled(mt_before)
    med = statistics.median
    return {"frames_per_s": 1.0 / med(t_frames), "ms_per_frame": med(t_frames) * 1e3, "ms_per_iteration": med(t_iter) * 1e3,
            "split_ms": {"update + new optimiser": med(t_update) * 1e3, "%d iterations" % iters: med(t_iter) * iters * 1e3,
                         "importance sweep": med(t_sweep) * 1e3},
            "frames_timed": len(t_frames), "final_loss": float(cur_loss.detach()),
            "autograd_nodes": ("C++ extension (lib/_shine_ext.so)" if ext == "1" else "Python (SHINE_TIER_A_EXT=0)") +
                              (", backward on the calling thread" if single_thread else ", engine's device thread"),
            "what": "the loop body of shine_incre.py:100-195 verbatim on the drop-in's classes and re-bound functions; eager launches, "
                    "one host synchronisation per phase (the driver's T0..T3)"}


if os.environ.get("TIER_A_SINGLE_THREAD_AUTOGRAD"):  # experiment: backward on the calling thread (no hand-over to the device thread)
    torch.autograd.set_multithreading_enabled(False)
CASES = (("maicity", 3, 4096), ("kitti", 3, 4096), ("maicity", 3, 1 << 16), ("kitti", 3, 1 << 16))
if os.environ.get("TIER_A_SMALL"):
    CASES = CASES[:2]
# the first case of a process reads the host's own warm-up (allocator, clocks: 0.53 against 0.31 ms for the same loop run second):
# one discarded pass of the first case in front
if __name__ != "__main__":
    CASES = ()
for case_i, (kind, lv, n) in enumerate(((CASES[0],) + tuple(CASES)) if CASES else ()):
    wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
    eik = bool(cfg.ekional_loss_on)
    g = torch.Generator(device="cuda").manual_seed(1)
    sigma = cfg.sigma_sigmoid

    def make_loop(patched, ext="1", single_thread=False):
        feats, mlp = list(octree.parameters()), list(dec.parameters())
        if patched:
            opt = optim.setup_optimizer(cfg, feats, mlp)
            bce, grad_fn = losses.sdf_bce_loss, losses.get_gradient
        else:
            groups = [{"params": mlp, "lr": cfg.lr, "weight_decay": cfg.weight_decay}] + \
                     [{"params": [feats[len(feats) - 1 - i]], "lr": cfg.lr} for i in range(len(feats))]
            opt = torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15)
            bce, grad_fn = losses._bce_composite, ref_get_gradient

        def loop():
            os.environ["SHINE_TIER_A_EXT"] = ext  # (read per call: _ext.module())
            torch.autograd.set_multithreading_enabled(not single_thread)
            autograd_ops.FUSE_WITH_COORD_GRAD = patched
            coord, sdf_label, weight = synth.draw_batch(wl.pool, n, g)
            if eik:
                coord.requires_grad_(True)
            feature = octree.query_feature(coord)
            sdf_pred = dec.sdf(feature)
            surface_mask = weight > 0
            if eik:
                gr = grad_fn(coord, sdf_pred) * sigma
            cur_loss = 0.
            weight = torch.abs(weight)
            cur_loss += bce(sdf_pred, sdf_label, sigma, weight, False, cfg.loss_reduction)
            if eik:
                cur_loss += cfg.weight_e * ((gr[surface_mask].norm(2, dim=-1) - 1.0) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            cur_loss.backward()
            opt.step()

        return loop

    o = StepOptions(sigma=sigma, ekional_loss_on=eik, weight_e=cfg.weight_e)
    adam_b = optim.setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())

    def tier_b():
        coord, label, weight = synth.draw_batch(wl.pool, n, g)
        fused_train_step(octree, dec, coord, label, weight, o)
        adam_b.step(zero_grad=True)

    loops = {"utils unpatched, Python nodes (r03)": make_loop(False, "0"), "dropin, Python nodes (r04)": make_loop(True, "0"),
             "dropin, C++ nodes, engine's device thread": make_loop(True, "1"),
             "dropin (r05): C++ nodes, backward on the calling thread": make_loop(True, "1", True),
             "tier B (fused step + fused Adam)": tier_b}
    times = {k: [] for k in loops}
    for fn in loops.values():
        for _ in range(5):
            fn()
    for rep in range(5):
        for name, fn in loops.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / 30 * 1e3)
    autograd_ops.FUSE_WITH_COORD_GRAD = False
    if case_i == 0:
        continue
    print(kind, "L%d" % lv, "N=%d" % n, "BCE+eikonal" if eik else "BCE",
          {k: "%.3f ms (min %.3f)" % (statistics.median(v), min(v)) for k, v in times.items()}, flush=True)

if __name__ == "__main__":  # config 4, Tier A: the incremental driver's frame, Python nodes then C++ nodes
    cfg_i = synth.make_config("ncd", device="cuda", tree_level_feat=3)
    fr = list(synth.make_frames(cfg_i, frames=10, beams=64, azimuths=900, seed=42, device="cuda"))
    for ext, st in (("0", False), ("1", False), ("1", True)):
        r = tier_a_incremental("cuda", fr, ext=ext, single_thread=st)
        print("ncd-incre tier A (%s): %.1f frames/s, %.2f ms/frame, %.3f ms/iteration, split %s" % (
            r["autograd_nodes"], r["frames_per_s"], r["ms_per_frame"], r["ms_per_iteration"],
            {k: round(v, 3) for k, v in r["split_ms"].items()}), flush=True)
