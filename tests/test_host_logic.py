"""CPU suite: host-side index logic around the HIP calls (no kernels are launched).

* chunk_partition — the chunks of cal_feature_importance (utils/incre_learning.py:27-31: pool[head:tail:down_rate]) as
  segments of a node-ordered permutation of the pool;
* canonical_order — SortedPool(canonical=True): samples of one node in ascending pool-index order, nodes left in place.
"""
import math

import pytest
import torch


@pytest.mark.parametrize("n,bs,down_rate", [(1, 4, 1), (1001, 100, 3), (700, 4096, 1), (1024, 64, 2), (8192, 4096, 2),
                                            (8193, 4096, 2), (50001, 2048, 5)])
def test_chunk_partition_reproduces_the_reference_slices(n, bs, down_rate):
    from shine_mapping_amd.incre_learning import chunk_partition

    g = torch.Generator().manual_seed(n)
    perm = torch.randperm(n, generator=g).to(torch.int32)  # any visiting order of the pool
    interval = bs * down_rate
    idx, begin = chunk_partition(perm, n, interval, down_rate)
    iter_n = math.ceil(n / interval)
    assert len(begin) == iter_n + 1 and begin[0] == 0 and idx.dtype == torch.int32
    pool = torch.arange(n)
    for c in range(iter_n):
        seg = idx[begin[c]:begin[c + 1]].long()
        want = pool[c * interval:min((c + 1) * interval, n):down_rate]  # the reference's slice
        assert torch.equal(torch.sort(perm.long()[seg]).values, want)
        assert bool((seg[1:] > seg[:-1]).all())  # visiting order is kept inside a chunk
    # what the stride skips sits behind the last chunk and is never read
    assert begin[-1] == sum(len(range(c * interval, min((c + 1) * interval, n), down_rate)) for c in range(iter_n))


def test_canonical_order_sorts_inside_nodes_only():
    from shine_mapping_amd.sampler import canonical_order

    g = torch.Generator().manual_seed(5)
    n, L = 5000, 3
    node = torch.sort(torch.randint(0, 300, (n,), generator=g)).values  # visiting order: by node
    slots = torch.stack([node // 25, node // 5, node], dim=1).to(torch.int32)
    slots[node % 7 == 0, 2] = -1  # some leaf-level misses: rows equal up to the coarser levels are interchangeable too
    base = torch.randperm(n, generator=g).to(torch.int32)
    order = canonical_order(base, slots)
    p2, s2 = base[order], slots[order]
    assert torch.equal(s2, slots)  # every run of identical rows stays where it was
    same = (s2[1:] == s2[:-1]).all(dim=1)
    assert bool((p2[1:][same] > p2[:-1][same]).all())  # ascending pool index inside a run
    assert torch.equal(torch.sort(p2).values, torch.sort(base).values)
    # two different within-node shuffles of the same pool end up identical
    shuffled = base.clone()
    for v in torch.unique(node)[:50]:
        m = torch.nonzero(node == v).flatten()
        shuffled[m] = base[m[torch.randperm(m.numel(), generator=g)]]
    assert not torch.equal(shuffled, base)
    assert torch.equal(shuffled[canonical_order(shuffled, slots)], p2)


def test_workload_cache_round_trips_the_built_workload(tmp_path, monkeypatch):
    """synth.build_workload with SHINE_WORKLOAD_CACHE set (tools/collect_profiles.sh: one process per counter group) must hand
    back the same octree tables, features and pool from the file as from the build."""
    from shine_mapping_amd import synth

    monkeypatch.setenv("SHINE_WORKLOAD_CACHE", str(tmp_path))
    kw = dict(frames=2, device="cpu", azimuths=60, tree_level_feat=3)
    a = synth.build_workload("maicity", **kw)
    files = list(tmp_path.iterdir())
    assert len(files) == 1
    b = synth.build_workload("maicity", **kw)  # loaded
    assert len(list(tmp_path.iterdir())) == 1
    assert all(torch.equal(x, y) for x, y in zip(a.octree.hier_features, b.octree.hier_features))
    assert torch.equal(a.pool.coord, b.pool.coord) and torch.equal(a.pool.weight, b.pool.weight)
    for s in range(3):
        assert (a.octree._node_keys[s] == b.octree._node_keys[s]).all() and (a.octree._node_ids[s] == b.octree._node_ids[s]).all()
    synth.build_workload("maicity", **dict(kw, frames=3))  # another key: another file
    assert len(list(tmp_path.iterdir())) == 2


def test_own_rows_exchange_with_synthetic_peers_multiplies_by_the_world():
    """dp.RowGatherReducer(synthetic_world=N) — bench.py's kitti-dp8-rank leg: without a process group the all-gather is N
    copies of this rank's own message, each unpacked and added.  The reduced bucket is then N times the rank's own gradients on
    the flagged rows (and the decoder tail), untouched rows stay zero, the flags are cleared for the next step."""
    from shine_mapping_amd.dp import RowGatherReducer

    g = torch.Generator().manual_seed(3)
    feats = [torch.nn.Parameter(torch.zeros(r, 8)) for r in (17, 41)]
    mlp = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    red = RowGatherReducer(feats, mlp, None, synthetic_world=4, capacity_rows=64)
    red._ensure_flat()
    rows = {0: [1, 5, 16], 1: [0, 7, 22, 40]}
    want = []
    for k, p in enumerate(feats):
        p.grad.zero_()
        for r in rows[k]:
            p.grad[r] = torch.randn(8, generator=g)
            red.flags[k][r] = 1
        want.append(p.grad.clone())
    for p in mlp:
        p.grad.copy_(torch.randn(p.shape, generator=g))
        want.append(p.grad.clone())
    red.exchange()
    assert red.world() == 4 and not red.overflowed()
    for p, w in zip(feats + mlp, want):
        assert torch.allclose(p.grad, 4.0 * w)
    for k, f in enumerate(red.flags):  # cleared, except the always-exchanged trash rows
        assert int(f[:-1].sum()) == 0 and int(f[-1]) == 1


def test_fused_adam_state_dict_has_torch_adams_layout():
    """ADVICE r04 (high), the host half: FusedAdam.state_dict() / load_state_dict() in torch.optim.Adam's layout — what the
    reference's save_checkpoint stores (utils/tools.py:200-213) — without a device: torch's Adam loads it, and a FusedAdam loads
    torch's (the GPU test test_fused_adam_state_dict_round_trips_through_torch_adam continues both and compares the steps)."""
    from shine_mapping_amd.optim import FusedAdam

    ps = [torch.nn.Parameter(torch.randn(4, 8)) for _ in range(3)]
    groups = [{"params": ps[:2], "lr": 0.01, "weight_decay": 1e-7}, {"params": [ps[2]], "lr": 0.005}]
    fa = FusedAdam(groups)
    assert fa.state_dict()["state"] == {}  # (torch creates a parameter's state at its first step)
    for p in ps:
        fa.state[p] = (torch.randn_like(p), torch.rand_like(p))
        fa._age[p] = 3
    fa.step_count = 3
    sd = fa.state_dict()
    ta = torch.optim.Adam([{"params": ps[:2], "lr": 0.3}, {"params": [ps[2]], "lr": 0.3}], betas=(0.5, 0.5), eps=1.0)
    assert sorted(sd["param_groups"][0]) == sorted(ta.state_dict()["param_groups"][0])
    ta.load_state_dict(sd)
    assert ta.param_groups[0]["lr"] == 0.01 and ta.param_groups[0]["betas"] == (0.9, 0.99) and ta.param_groups[1]["lr"] == 0.005
    assert float(ta.state[ps[0]]["step"]) == 3.0 and torch.equal(ta.state[ps[2]]["exp_avg"], fa.state[ps[2]][0])
    fb = FusedAdam(groups, betas=(0.1, 0.1), eps=1.0)
    fb.load_state_dict(ta.state_dict())
    assert fb.betas == (0.9, 0.99) and fb.eps == 1e-15 and fb.step_count == 3 and fb._age[ps[1]] == 3
    assert torch.equal(fb.state[ps[1]][1], fa.state[ps[1]][1]) and fb.param_groups[1]["lr"] == 0.005
    with pytest.raises(ValueError):
        FusedAdam(groups[:1]).load_state_dict(sd)
