"""GPU suite: the data-parallel step of bench.py with TWO real ranks on the HIP path.

The box has one GPU, so both rank processes use cuda:0 and the collective is gloo on host copies (RCCL refuses two ranks
on one device) — everything else is the code `bench.py --gpus N` runs per rank: the rank's contiguous slice of ONE global
sorted draw (shine_sample_sorted_slice), the fused step with the GLOBAL normalisers, the 8-byte surface-count
all-reduce, then either the dense flat-bucket all-reduce or the touched-row exchange (device index / pack / unpack
kernels around the collective).  The reduced gradients and the summed loss must equal ONE process running the whole
global batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from shine_mapping_amd import StepOptions, fused_train_step, synth
from shine_mapping_amd import dp as shine_dp
from shine_mapping_amd.sampler import SortedPool

kind, exchange, points, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
single = world == 1
if not single:
    dist.init_process_group("gloo", rank=rank, world_size=world)


class HostStaged:
    """torch.distributed's all_reduce for CUDA tensors through a host copy (two ranks share one GPU here)."""
    ReduceOp = dist.ReduceOp

    @staticmethod
    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        c = t.detach().cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)

    @staticmethod
    def get_world_size(group=None):
        return dist.get_world_size(group)

    @staticmethod
    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        c = inp.detach().cpu()
        parts = [torch.empty_like(c) for _ in range(dist.get_world_size(group))]
        if not async_op:
            dist.all_gather(parts, c, group=group)
            out.copy_(torch.cat(parts))
            return None
        work = dist.all_gather(parts, c, group=group, async_op=True)  # in flight under whatever the caller launches next

        class Pending:
            def wait(self):
                work.wait()
                out.copy_(torch.cat(parts))

        return Pending()


torch.cuda.set_device(0)
wl = synth.build_workload(kind, frames=8, device="cuda", seed=42, tree_level_feat=3, azimuths=300)
cfg, octree, decoder, pool = wl.cfg, wl.octree, wl.decoder, wl.pool
n_global = points * world
opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction=cfg.loss_reduction, ekional_loss_on=cfg.ekional_loss_on,
                   weight_e=cfg.weight_e, n_global=n_global)
params = list(octree.hier_features) + decoder.fused_params()
for p in params:
    p.grad = torch.zeros_like(p)
micro = 2 if exchange == "gather-micro2" else 1
if micro > 1:  # bench.py --micro-batches 2: the (asynchronous) all-gather of micro-batch k under the fused kernel of k + 1
    exchange = "gather"
Reducer = shine_dp.RowGatherReducer if exchange == "gather" else shine_dp.TouchedRowReducer
reducer = Reducer(list(octree.hier_features), decoder.fused_params(), None if single else HostStaged,
                  **({"async_op": True} if micro > 1 else {}))
octree._require_tables(with_ranks=True)
spool = SortedPool(octree, pool.coord, pool.sdf_label, pool.weight, seed=1000, canonical=True)
flags = None
if exchange == "touched":
    flags = shine_dp.mark_touched(octree, spool, spool.draw(8))
    for f in flags:
        f.zero_()
else:
    spool.draw(8)  # keep the draw counter in step with the touched variant
results = []
for it in range(3):
    idx = spool.draw(points, zero=reducer.flat, n_global=n_global, slice_begin=rank * points)
    n_surf = None
    if opts.ekional_loss_on:
        n_surf = (spool.weight[idx.long()] > 0).sum()
        reducer.all_reduce_scalar(n_surf)
    if micro > 1:  # bench.py run_micro: contiguous sub-slices, exchange after each, everything added back after the last
        loss, preds = None, []
        for k in range(micro):
            a_, b_ = k * points // micro, (k + 1) * points // micro
            l_, p_, _ = fused_train_step(octree, decoder, None, None, None, opts, n_surf=n_surf, pool=spool, idx=idx[a_:b_],
                                         touched=reducer.flags)
            loss = l_.detach().clone() if loss is None else loss + l_.detach()
            preds.append(p_.detach().clone())
            reducer.exchange(finish=k == micro - 1)
        pred = torch.cat(preds)
        assert not reducer.overflowed()
    else:
        loss, pred, _ = fused_train_step(octree, decoder, None, None, None, opts, n_surf=n_surf, pool=spool, idx=idx,
                                         touched=reducer.flags if exchange == "gather" else None)
        loss = loss.detach().clone()
    if micro > 1:
        pass
    elif exchange == "gather":  # own rows (flagged by the step itself) -> message -> all-gather -> added back
        reducer.exchange()
        assert not reducer.overflowed()
    elif exchange == "touched":
        shine_dp.mark_touched(octree, spool, idx, flags)
        reducer.or_reduce_flags(flags)
        reducer.all_reduce_touched(flags)
        assert all(int(f.sum()) == 0 for f in flags)  # cleared for the next step
    else:
        reducer.all_reduce_grads()
    reducer.all_reduce_scalar(loss)
    results.append(dict(loss=loss.cpu(), idx=idx.cpu(), pred=pred.detach().cpu().clone(), grads=[p.grad.detach().cpu().clone() for p in params],
                        rows=reducer.last_rows, bytes=reducer.last_bytes))
torch.cuda.synchronize()
torch.save(results, out + ".rank%d" % rank)
if not single:
    dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, kind, exchange, points, world, tag):
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / tag)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, kind, exchange, str(points), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, logs[rank][-3000:])
    return [torch.load(out + ".rank%d" % r, weights_only=False) for r in range(world)]


@pytest.mark.parametrize("kind,exchange", [("maicity", "dense"), ("maicity", "touched"), ("kitti", "dense"),
                                           ("kitti", "touched"), ("maicity", "gather"), ("kitti", "gather"),
                                           ("kitti", "gather-micro2")])
def test_two_ranks_match_one_process(kind, exchange, tmp_path):
    points = 8192 + 40  # per rank; ragged against the 16-point tiles and the sampler's 1024-draw blocks
    two = _run(tmp_path, kind, exchange, points, 2, "two")
    one = _run(tmp_path, kind, "dense", 2 * points, 1, "one")[0]
    for it in range(3):
        r0, r1, ref = two[0][it], two[1][it], one[it]
        # the two slices are the halves of the single process's global sorted draw
        assert torch.equal(torch.cat([r0["idx"], r1["idx"]]), ref["idx"])
        pred2 = torch.cat([r0["pred"], r1["pred"]])
        bad = torch.nonzero(pred2 != ref["pred"]).flatten()  # same math per point: bit-equal
        assert bad.numel() == 0, (it, int(bad.numel()), bad[:12].tolist(), bad[-4:].tolist(),
                                  float((pred2 - ref["pred"]).abs().max()), ref["idx"][bad[:12]].tolist())
        assert torch.allclose(r0["loss"], ref["loss"], rtol=2e-5, atol=1e-7), (it, r0["loss"], ref["loss"])
        assert torch.equal(r0["loss"], r1["loss"])
        for k, (a, b, c) in enumerate(zip(r0["grads"], r1["grads"], ref["grads"])):
            assert torch.equal(a, b), "replicas diverge on tensor %d" % k  # every rank holds the same reduced grads
            scale = float(c.abs().max()) + 1e-30
            assert float((a - c).abs().max()) <= 1e-4 * scale + 1e-9, (it, k, float((a - c).abs().max()), scale)
        if exchange == "touched":
            dense_rows = sum(int(g.shape[0]) - 1 for g in r0["grads"][:3])
            assert 0 < r0["rows"] <= dense_rows and r0["bytes"] == r1["bytes"]
