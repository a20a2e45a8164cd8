#!/usr/bin/env python
"""The fused kernel at the reference's batch size (N = 4096) by input mode: pool mode (perm -> pool sample -> slot -> ids -> rows: the
product's graphed iteration) against a MATERIALISED batch (coord / label / slots gathered beforehand: one dependent load less in
front of the first tile).  Kernel time only (no reduction launch), HIP events around 200 back-to-back launches.
    python tools/small_batch_modes.py [maicity|ncd] [levels]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd.sampler import SortedPool

kind = sys.argv[1] if len(sys.argv) > 1 else "maicity"
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = synth.build_workload(kind, frames=30, device="cuda", seed=42, tree_level_feat=lv, azimuths=450)
octree, dec, cfg = wl.octree, wl.decoder, wl.cfg
for p in list(octree.hier_features) + dec.fused_params():
    p.grad = torch.zeros_like(p)
octree._require_tables(with_ranks=True)
sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight)
n = 4096
idx = sp.draw(n)
c, l, w = (t.contiguous() for t in sp.get_batch(idx))
slots_b = sp.slots[idx.long()].contiguous()
eik = bool(cfg.ekional_loss_on)
ns = (w > 0).sum() if eik else None
o = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=eik, weight_e=cfg.weight_e, kernel_variant=0x2000)


def pool_mode():
    fused_train_step(octree, dec, None, None, None, o, n_surf=ns, pool=sp, idx=idx)


def materialised():
    fused_train_step(octree, dec, c, l, w, o, n_surf=ns, slots=slots_b)


res = {}
for rep in range(4):
    for name, fn in (("pool mode", pool_mode), ("materialised batch", materialised)):
        for _ in range(20):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 200 * 1e3)
print(kind, "L%d" % lv, "N=%d" % n, {k: ["%.2f" % v for v in vs] for k, vs in res.items()}, "us per launch (graph of 50)")
