#!/usr/bin/env python
"""First-contact check of kernel_variant 5 (shine_step_v5.hip) against kernel_variant 4 (shine_step_v3.hip) on planned
batches of the goldens: loss / pred / every gradient, BCE and eikonal.  Run under `timeout`: a broken hand-off shows up as
NaN loss (bounded spins), not as a hang."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden, product_from_golden
from shine_mapping_amd import StepOptions, dp, fused_train_step

bad = 0
for name in ("maicity_bce_L4", "maicity_bce_L3", "kitti_eik_L3", "ncd_reg_L3"):
    fx = load_golden(name)
    cfg, octree, dec = product_from_golden(fx)
    eik = bool(fx["cfg"].get("ekional_loss_on", False))
    for reps in (1, 9):
        c = fx["coord"].cuda().repeat(reps, 1).contiguous()
        l = fx["sdf_label"].cuda().repeat(reps).contiguous()
        w = fx["weight"].cuda().repeat(reps).contiguous()
        if reps > 1:
            torch.manual_seed(reps)
            c = (c + 1e-5 * torch.randn_like(c)).contiguous()
        perm, slots = dp.plan_batch(octree, c)
        params = list(octree.hier_features) + dec.fused_params()
        outs = {}
        for v in (4, 5):
            for p in params:
                p.grad = None
            o = StepOptions(sigma=fx["sigma"], ekional_loss_on=eik, weight_e=fx["cfg"].get("weight_e", 0.1),
                            loss_reduction=fx["cfg"].get("loss_reduction", "mean"), kernel_variant=v)
            loss, pred, g = fused_train_step(octree, dec, c, l, w, o, want_grad_x=True, perm=perm, slots=slots)
            torch.cuda.synchronize()
            outs[v] = (float(loss), pred.double().clone(), None if g is None else g.double().clone(),
                       [p.grad.double().clone() for p in params])
        a, b = outs[4], outs[5]
        e_pred = float((a[1] - b[1]).abs().max())
        e_g = 0.0 if a[2] is None else float((a[2] - b[2]).abs().max() / a[2].abs().max().clamp_min(1e-30))
        e_gr = max(float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) for x, y in zip(a[3], b[3]))
        ok = abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(a[0])) and e_pred <= 1e-5 and e_g <= 1e-4 and e_gr <= 1e-4
        bad += 0 if ok else 1
        print("%-16s n=%6d  loss v3 %.8g v5 %.8g | pred %.2e | g %.2e | grads %.2e  %s" % (
            name, c.shape[0], a[0], b[0], e_pred, e_g, e_gr, "ok" if ok else "MISMATCH"))
print("v5_check:", "ALL OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
