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


if os.environ.get("TIER_A_SINGLE_THREAD_AUTOGRAD"):  # experiment: backward on the calling thread (no hand-over to the device thread)
    torch.autograd.set_multithreading_enabled(False)
CASES = (("maicity", 3, 4096), ("kitti", 3, 4096), ("maicity", 3, 1 << 16), ("kitti", 3, 1 << 16))
if os.environ.get("TIER_A_SMALL"):
    CASES = CASES[:2]
# the first case of a process reads the host's own warm-up (allocator, clocks: 0.53 against 0.31 ms for the same loop run second):
# one discarded pass of the first case in front
for case_i, (kind, lv, n) in enumerate((CASES[0],) + tuple(CASES)):
    wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
    eik = bool(cfg.ekional_loss_on)
    g = torch.Generator(device="cuda").manual_seed(1)
    sigma = cfg.sigma_sigmoid

    def make_loop(patched, ext="1"):
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
             "dropin, C++ nodes (r05)": make_loop(True, "1"), "tier B (fused step + fused Adam)": tier_b}
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
