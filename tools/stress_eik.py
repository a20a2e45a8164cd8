"""Stress: repeated pool / planned / v0 eikonal steps on the same points; reports any run whose outputs differ."""
import sys, torch
sys.path.insert(0, '.')
from shine_mapping_amd import StepOptions, dp, fused_train_step, synth
from shine_mapping_amd.sampler import SortedPool
kind, levels, n = "kitti", 3, (1 << 17) + 1
wl = synth.build_workload(kind, frames=8, device="cuda", seed=21, tree_level_feat=levels, azimuths=300)
octree, dec, cfg = wl.octree, wl.decoder.cuda(), wl.cfg
with torch.no_grad():
    for p in octree.hier_features:
        p.mul_(5.0)
octree._require_tables(with_ranks=True)
sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=2)
idx = sp.draw(n)
params = list(octree.hier_features) + dec.fused_params()
c, l, w = (t.contiguous() for t in sp.get_batch(idx))
perm, slots = dp.plan_batch(octree, c)
def run(mode, variant=0):
    for p in params: p.grad = None
    opts = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, kernel_variant=variant)
    junk = torch.full((n, 3), float('nan'), device='cuda')  # poison what the allocator hands out next
    del junk
    if mode == 'pool':
        loss, pred, g = fused_train_step(octree, dec, None, None, None, opts, want_grad_x=True, pool=sp, idx=idx)
    elif mode == 'planned':
        loss, pred, g = fused_train_step(octree, dec, c, l, w, opts, want_grad_x=True, perm=perm, slots=slots)
    else:
        loss, pred, g = fused_train_step(octree, dec, c, l, w, opts, want_grad_x=True)
    torch.cuda.synchronize()
    return float(loss), pred.clone(), g.clone(), [p.grad.clone() for p in params]
ref = run('plain', 1)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    perm, slots = dp.plan_batch(octree, c)  # a fresh plan every round: it must always be a permutation
    cnt = torch.bincount(perm.long(), minlength=n)
    if int((cnt != 1).sum()):
        print("PLAN NOT A PERMUTATION at it", it, int((cnt != 1).sum()))
    sp2 = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=2)
    if not (torch.equal(sp2.slots, sp.slots) and torch.equal(sp2.coord, sp.coord)):
        print("POOL PLAN DIFFERS at it", it)
    for mode, var in (('pool', 0), ('planned', 0), ('plain', 2), ('pool', 2)):
        lo, pr, g, gr = run(mode, var)
        dg = (g - ref[2]).abs().max(1).values
        dp_ = (pr - ref[1]).abs()
        nb = int((~(dg <= 1e-6)).sum()); npb = int((~(dp_ <= 2e-5)).sum())
        gerr = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(gr, ref[3]))
        if nb or npb or not (gerr <= 1e-4) or abs(lo - ref[0]) > 1e-5:
            bad += 1
            ii = (~(dg <= 1e-6)).nonzero().flatten()
            print("MISMATCH it %d %s/%d: loss %.9f vs %.9f, %d bad g rows, %d bad pred, grad err %.2e; rows %s nan? %s" % (
                it, mode, var, lo, ref[0], nb, npb, gerr, ii[:8].tolist(), bool(torch.isnan(g).any())))
            if nb:
                i = int(ii[0]); print("   g", g[i].tolist(), "ref", ref[2][i].tolist())
print("done, mismatching runs:", bad)
