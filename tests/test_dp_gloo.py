"""CPU suite: the data-parallel path with world_size 2 over gloo.

The N>1 design: replicated tables + decoder, each rank runs the step on its shard of the global batch with
the GLOBAL normalisers (1/N_global for 'mean', global surface count for the eikonal mean), then one flat
all-reduce(sum) of the dense grads.  Here the per-rank compute is the CPU oracle (tests may use it as the
stand-in; the product path is HIP-only) — what is under test is shine_mapping_amd.dp.GradReducer and the
normaliser algebra: reduced shard grads == single-process full-batch grads."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, oracle_from_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded_step(oct_, mlp, c, l, w, cfg, n_global, n_surf_global, reset=True):
    """The oracle's step with GLOBAL normalisers: what each rank's fused kernel computes under DP."""
    from oracle import shine_oracle as so

    if reset:
        oct_.zero_grad()
        mlp.zero_grad()
    sig = so.sigma_sigmoid(cfg)
    eik = bool(cfg.ekional_loss_on)
    coord = c.detach().clone().requires_grad_(eik)
    pred = mlp.sdf(oct_.query_feature(coord))
    loss = so.sdf_bce_loss(pred, l, sig, "sum") / n_global
    if eik:
        g = so.coord_gradient(coord, pred) * sig
        e = ((1.0 - g[w > 0].norm(2, dim=-1)) ** 2).sum() / n_surf_global
        loss = loss + cfg.weight_e * e
    loss.backward()
    return loss.detach()


def _worker(rank, world, port, name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from shine_mapping_amd.dp import GradReducer

    fx = load_golden(name)
    cfg, oct_, mlp = oracle_from_golden(fx)
    n = fx["coord"].shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world
    c, l, w = fx["coord"][lo:hi], fx["sdf_label"][lo:hi], fx["weight"][lo:hi]
    params = list(oct_.hier_features) + mlp.params()
    reducer = GradReducer(params, dist)
    n_surf = (w > 0).sum()
    reducer.all_reduce_scalar(n_surf)
    loss = _sharded_step(oct_, mlp, c, l, w, cfg, n, int(n_surf))
    reducer.all_reduce_grads()
    dist.all_reduce(loss)
    # second iteration re-uses the flat bucket (grads are views into it)
    for p in params:
        p.grad.zero_()
    _sharded_step(oct_, mlp, c, l, w, cfg, n, int(n_surf), reset=False)
    for p in params:  # autograd accumulates in place into the bucket views
        assert p.grad.data_ptr() >= reducer._flat.data_ptr()
    reducer.all_reduce_grads()
    if rank == 0:
        torch.save(dict(loss=loss, grads=[p.grad.clone() for p in params]), os.path.join(out_dir, "dp.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["maicity_bce_L3", "kitti_eik_L3"])
def test_two_rank_gloo_matches_single_process(name, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "dp.pt"), weights_only=False)
    fx = load_golden(name)
    ref = fx["out"]
    refs = list(ref["feat_grads"]) + list(ref["mlp_grads"])
    assert abs(float(got["loss"]) - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"])))
    for g, r in zip(got["grads"], refs):
        assert float((g - r).abs().max()) <= 1e-5 * max(float(r.abs().max()), 1e-30)


def _worker_touched(rank, world, port, name, out_dir):
    """§8(e)'s partition: ONE global batch in node order, rank r takes the r-th contiguous slice; every rank marks the
    rows of the global batch itself (no collective) and only those rows travel."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from shine_mapping_amd.dp import TouchedRowReducer

    fx = load_golden(name)
    cfg, oct_, mlp = oracle_from_golden(fx)
    n = fx["coord"].shape[0]
    # the global batch, ordered by leaf-level node (what the sorted pool draw produces), known to every rank
    hidx = oct_.get_indices(fx["coord"])
    order = torch.argsort(hidx[0][:, 0], stable=True)
    gc, gl, gw = fx["coord"][order], fx["sdf_label"][order], fx["weight"][order]
    L = len(oct_.hier_features)
    flags = []
    for s in range(L):
        f = torch.zeros(oct_.hier_features[s].shape[0], dtype=torch.uint8)
        u = hidx[L - 1 - s].flatten().unique()
        f[u[u >= 0]] = 1
        flags.append(f)
    lo, hi = rank * n // world, (rank + 1) * n // world
    reducer = TouchedRowReducer(list(oct_.hier_features), mlp.params(), dist)
    n_surf = int((gw > 0).sum())  # from the common draw: no scalar all-reduce
    loss = _sharded_step(oct_, mlp, gc[lo:hi], gl[lo:hi], gw[lo:hi], cfg, n, n_surf)
    local = [p.grad.clone() for p in reducer.params]
    reducer.all_reduce_touched([f.clone() for f in flags])
    sparse = [p.grad.clone() for p in reducer.params]
    for p, g in zip(reducer.params, local):  # same local grads through the dense bucket
        p.grad.copy_(g)
    reducer.all_reduce_grads()
    dist.all_reduce(loss)
    if rank == 0:
        torch.save(dict(loss=loss, sparse=sparse, dense=[p.grad.clone() for p in reducer.params],
                        rows=reducer.last_rows, bytes=reducer.last_bytes, dense_bytes=reducer.dense_bytes()),
                   os.path.join(out_dir, "dp_touched.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["maicity_bce_L3", "kitti_eik_L3"])
def test_touched_row_exchange_equals_dense_all_reduce(name, tmp_path):
    world = 2
    mp.spawn(_worker_touched, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "dp_touched.pt"), weights_only=False)
    fx = load_golden(name)
    ref = fx["out"]
    refs = list(ref["feat_grads"]) + list(ref["mlp_grads"])
    assert abs(float(got["loss"]) - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"])))
    for s, d, r in zip(got["sparse"], got["dense"], refs):
        assert torch.equal(s, d), "touched-row exchange differs from the dense all-reduce"
        assert float((s - r).abs().max()) <= 1e-5 * max(float(r.abs().max()), 1e-30)
    assert 0 < got["bytes"] < got["dense_bytes"]  # 4096 points touch a fraction of the fixture's rows


def _worker_gather(rank, world, port, name, out_dir):
    """The own-rows all-gather exchange (dp.RowGatherReducer): every rank flags the rows ITS slice touches, moves them into
    a fixed-capacity message and the ranks all-gather — once per step, and once per MICRO-batch of a step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from shine_mapping_amd.dp import RowGatherReducer

    fx = load_golden(name)
    cfg, oct_, mlp = oracle_from_golden(fx)
    n = fx["coord"].shape[0]
    hidx = oct_.get_indices(fx["coord"])
    order = torch.argsort(hidx[0][:, 0], stable=True)
    gc, gl, gw = fx["coord"][order], fx["sdf_label"][order], fx["weight"][order]
    L = len(oct_.hier_features)
    lo, hi = rank * n // world, (rank + 1) * n // world
    reducer = RowGatherReducer(list(oct_.hier_features), mlp.params(), dist, async_op=True)  # (finish() waits)
    n_surf = int((gw > 0).sum())

    def mark(a, b):  # what the fused step's touched pass does for points [a, b) of the global batch
        sub = oct_.get_indices(gc[a:b])
        for s in range(L):
            u = sub[L - 1 - s].flatten().unique()
            reducer.flags[s][u[u >= 0]] = 1

    # (1) one exchange per step
    loss = _sharded_step(oct_, mlp, gc[lo:hi], gl[lo:hi], gw[lo:hi], cfg, n, n_surf)
    reducer._ensure_flat()
    local = [p.grad.clone() for p in reducer.params]
    mark(lo, hi)
    reducer.exchange()
    whole = [p.grad.clone() for p in reducer.params]
    cap_measured = reducer.capacity
    # (2) the same step as two micro-batches, each exchanged on its own, added back at the end
    for p in reducer.params:
        p.grad.zero_()
    mid = (lo + hi) // 2
    _sharded_step(oct_, mlp, gc[lo:mid], gl[lo:mid], gw[lo:mid], cfg, n, n_surf, reset=False)
    mark(lo, mid)
    reducer.exchange(finish=False)
    assert all(float(p.grad.abs().max()) == 0.0 for p in reducer.params), "the pack moves the rows (and the tail) out"
    _sharded_step(oct_, mlp, gc[mid:hi], gl[mid:hi], gw[mid:hi], cfg, n, n_surf, reset=False)
    mark(mid, hi)
    reducer.exchange(finish=True)
    micro = [p.grad.clone() for p in reducer.params]
    ok = not reducer.overflowed()
    # (3) a message too small for the rows: reported, not silent
    small = RowGatherReducer(list(oct_.hier_features), mlp.params(), dist, capacity_rows=8)
    for p, g in zip(small.params, local):
        p.grad.copy_(g)
    sub = oct_.get_indices(gc[lo:hi])
    for s in range(L):
        u = sub[L - 1 - s].flatten().unique()
        small.flags[s][u[u >= 0]] = 1
    small.exchange()
    overflow_seen = small.overflowed()
    # the dense all-reduce of the same local grads
    for p, g in zip(reducer.params, local):
        p.grad.copy_(g)
    reducer.all_reduce_grads()
    dist.all_reduce(loss)
    if rank == 0:
        torch.save(dict(loss=loss, whole=whole, micro=micro, dense=[p.grad.clone() for p in reducer.params], ok=ok,
                        overflow_seen=overflow_seen, bytes=reducer.last_bytes, dense_bytes=reducer.dense_bytes(),
                        cap=cap_measured, flags_left=int(reducer._flags_flat.sum()), levels=L),
                   os.path.join(out_dir, "dp_gather.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["maicity_bce_L3", "kitti_eik_L3"])
def test_own_rows_all_gather_equals_dense_all_reduce(name, tmp_path):
    world = 2
    mp.spawn(_worker_gather, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "dp_gather.pt"), weights_only=False)
    ref = load_golden(name)["out"]
    refs = list(ref["feat_grads"]) + list(ref["mlp_grads"])
    assert abs(float(got["loss"]) - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"])))
    for w, m, d, r in zip(got["whole"], got["micro"], got["dense"], refs):
        scale = max(float(r.abs().max()), 1e-30)
        assert float((w - d).abs().max()) <= 1e-6 * scale, "own-rows all-gather differs from the dense all-reduce"
        assert float((m - d).abs().max()) <= 1e-5 * scale, "two exchanged micro-batches differ from the dense all-reduce"
        assert float((w - r).abs().max()) <= 1e-5 * scale
    assert got["ok"] and got["overflow_seen"]
    assert got["flags_left"] == got["levels"]  # only the trash rows stay flagged
    assert 0 < got["bytes"] and got["cap"] % 4 == 0
