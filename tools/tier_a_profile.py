#!/usr/bin/env python
"""Where the HOST time of a Tier A iteration goes (cProfile over the dropin loop of tools/tier_a_bench.py).
    python tools/tier_a_profile.py [maicity|kitti] [iterations]"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import autograd_ops, losses, optim, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "maicity"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=3)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio = True, 1e-15, 1.0
eik = bool(cfg.ekional_loss_on)
g = torch.Generator(device="cuda").manual_seed(1)
sigma = cfg.sigma_sigmoid
opt = optim.setup_optimizer(cfg, list(octree.parameters()), list(dec.parameters()))
autograd_ops.FUSE_WITH_COORD_GRAD = True


def loop():
    coord, sdf_label, weight = synth.draw_batch(wl.pool, 4096, g)
    if eik:
        coord.requires_grad_(True)
    feature = octree.query_feature(coord)
    sdf_pred = dec.sdf(feature)
    surface_mask = weight > 0
    if eik:
        gr = losses.get_gradient(coord, sdf_pred) * sigma
    cur_loss = 0.
    weight = torch.abs(weight)
    cur_loss += losses.sdf_bce_loss(sdf_pred, sdf_label, sigma, weight, False, cfg.loss_reduction)
    if eik:
        cur_loss += cfg.weight_e * ((gr[surface_mask].norm(2, dim=-1) - 1.0) ** 2).mean()
    opt.zero_grad(set_to_none=True)
    cur_loss.backward()
    opt.step()


for _ in range(20):
    loop()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    loop()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
print("== %s, %d iterations: by internal time (divide by %d for per-iteration)" % (kind, iters, iters))
st.sort_stats("tottime").print_stats(45)
print("== by cumulative time, this package only")
st.sort_stats("cumulative").print_stats("shine_mapping_amd|tier_a_profile", 45)
