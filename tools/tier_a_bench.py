#!/usr/bin/env python
"""Tier A (strict drop-in) iteration time: the reference's loop body (shine_batch.py:115-210) on this package's classes —
query_feature (OctreeInterp) -> sdf (FusedMLP) -> [get_gradient] -> sdf_bce_loss [+ eikonal] -> backward -> torch Adam —
next to the fused Tier-B step on the same unordered batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.losses import sdf_bce_loss


def get_gradient(inputs, outputs):  # utils/tools.py:175-185
    d = torch.ones_like(outputs, requires_grad=False)
    return torch.autograd.grad(outputs=outputs, inputs=inputs, grad_outputs=d, create_graph=True, retain_graph=True,
                               only_inputs=True)[0]


for kind, lv, n in (("maicity", 3, 4096), ("maicity", 3, 1 << 16), ("kitti", 3, 4096), ("kitti", 3, 1 << 16)):
    wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=lv)
    octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
    eik = bool(cfg.ekional_loss_on)
    params = list(octree.parameters()) + list(dec.parameters())
    opt = torch.optim.Adam([p for p in params if p.requires_grad], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device="cuda").manual_seed(1)
    sigma = cfg.sigma_sigmoid

    def tier_a():
        coord, label, weight = synth.draw_batch(wl.pool, n, g)
        if eik:
            coord.requires_grad_(True)
        feature = octree.query_feature(coord)
        pred = dec.sdf(feature)
        loss = sdf_bce_loss(pred, label, sigma, None, False, cfg.loss_reduction)
        if eik:
            gr = get_gradient(coord, pred) * sigma
            loss = loss + cfg.weight_e * ((1.0 - gr[weight > 0].norm(2, dim=-1)) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    o = StepOptions(sigma=sigma, ekional_loss_on=eik, weight_e=cfg.weight_e)

    def tier_b():
        coord, label, weight = synth.draw_batch(wl.pool, n, g)
        for p in params:
            if p.grad is not None:
                p.grad.zero_()
        fused_train_step(octree, dec, coord, label, weight, o)
        opt.step()

    out = {}
    for name, fn in (("tier A", tier_a), ("tier B (fused, torch Adam)", tier_b)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 30 * 1e3
    print(kind, "L%d" % lv, "N=%d" % n, "eikonal" if eik else "BCE", {k: "%.2f ms" % v for k, v in out.items()})
