"""cal_feature_importance — utils/incre_learning.py:8-40 on the fused HIP path.

Per chunk of the frame's pool the reference runs query + decode + BCE + backward and then
``importance_weight[i] += hier_features[i].grad.abs(); grad.zero_(); importance_weight[i][-1] *= 0``.
Here the forward+backward of up to 64 chunks is ONE launch of the fused step with the decoder frozen (every chunk with its own
gradient tables: a chunk's gradient is summed before the abs) and one more launch folds them into importance_weight
(csrc/shine_sweep.hip).
"""
import ctypes as C
import math

import torch

from . import _lib
from .dp import plan_batch
from .ops import StepOptions, _dense_grad

# scratch of the sweep (the chunks' private gradient tables and row flags): zero on entry, left zero by the call, so ONE buffer per
# device is kept and re-used from frame to frame.  It is sized for the map at hand (+ 1/8 so that a growing map does not
# re-allocate every frame), dropped when the map needs less than half of it, capped by SCRATCH_BUDGET_BYTES and by a quarter of
# the device memory that is free when it is (re)allocated — fewer chunks per launch then, not more memory — and
# release_scratch() hands it back.
_SCRATCH = {}
_BUDGET = {}
SCRATCH_BUDGET_BYTES = 2 << 30


def _scratch_budget(dev) -> int:
    b = _BUDGET.get(dev)
    if b is None:  # (asked of the driver once per (re)allocation, not once per frame)
        held = _SCRATCH.get(dev)
        free = torch.cuda.mem_get_info(dev)[0] + (held.numel() if held is not None else 0)
        b = _BUDGET[dev] = int(max(64 << 20, min(SCRATCH_BUDGET_BYTES, free // 4)))
    return b


def _zero_scratch(dev, nbytes):
    buf = _SCRATCH.get(dev)
    if buf is None or buf.numel() < nbytes or buf.numel() > 2 * nbytes + (16 << 20):
        _SCRATCH.pop(dev, None)
        buf = None  # (the old buffer goes back to the allocator before the new one is asked for)
        buf = _SCRATCH[dev] = torch.zeros(nbytes + nbytes // 8, dtype=torch.uint8, device=dev)
        _BUDGET.pop(dev, None)
    return buf


def release_scratch(dev=None):
    """Give the sweep's scratch buffer(s) back to torch's allocator (kept otherwise for the life of the process)."""
    if dev is None:
        _SCRATCH.clear()
        _BUDGET.clear()
    else:
        _SCRATCH.pop(torch.device(dev), None)
        _BUDGET.pop(torch.device(dev), None)


def chunk_partition(perm, sample_count, batch_interval, down_rate):
    """(The torch form of shine_importance_chunks, kept as its cross-check in the tests.)  The reference's chunks
    (pool[head:tail:down_rate] for head = n * batch_interval, utils/incre_learning.py:27-31) as segments of node-ordered positions: perm[j] is the pool index of sorted position j.  Returns (idx int32 [kept],
    begin list[iter_n + 1]): chunk n = idx[begin[n]:begin[n+1]], ascending (= node order) inside a chunk."""
    iter_n = math.ceil(sample_count / batch_interval)
    p = perm.long()
    off = p % batch_interval
    # the samples the stride skips go behind the last chunk
    key = torch.where(off % down_rate == 0, p // batch_interval, torch.full_like(p, iter_n))
    idx = torch.argsort(key, stable=True).to(torch.int32)
    begin = [0]
    for n in range(iter_n):
        head, tail = n * batch_interval, min((n + 1) * batch_interval, sample_count)
        begin.append(begin[-1] + (tail - head + down_rate - 1) // down_rate)
    return idx, begin


def cal_feature_importance(data, octree, mlp, sigma, bs, down_rate=1, loss_reduction="mean", loss_weight_on=False, pool=None):
    """Same signature as the reference; `data` needs .coord_pool and .sdf_label_pool (utils/incre_learning.py:14-26).
    `pool` (extension): the frame's sampler.SortedPool built from the SAME data.coord_pool / sdf_label_pool — its plan (node
    order, hash slots, node-ordered copies) is re-used instead of planning the pool a second time.

    The reference walks the pool in chunks [n*bs*down_rate, (n+1)*bs*down_rate) taking every down_rate-th sample
    (:27-31); a chunk's gradient is summed before the abs (:36-38), so chunk MEMBERSHIP is part of the result, the order
    inside a chunk is not.  Here the whole pool is planned once (node order + hash slots), the node-ordered samples are
    partitioned by chunk with one stable device sort, and shine_importance_sweep runs the chunk loop (fused step with the
    decoder frozen + epilogue) on the other side of the ABI: no per-chunk Python, slicing or planning."""
    # loss_weight_on has no effect here, as in the reference: it passes weight=None (utils/incre_learning.py:32), and
    # nn.BCEWithLogitsLoss(weight=None) is the unweighted loss
    dev = octree.hier_features[0].device
    coord_pool = data.coord_pool.to(dev)
    label_pool = data.sdf_label_pool.to(dev, torch.float32)
    sample_count = coord_pool.shape[0]
    batch_interval = bs * down_rate
    iter_n = math.ceil(sample_count / batch_interval)
    if iter_n == 0:
        return
    t = octree._require_tables(with_ranks=True, probe=False)  # (the sweep reads memoised slots; a plan below records itself)
    if pool is not None:
        if pool.size != sample_count or pool.tables_epoch != octree._tables_epoch or pool.coord.device != dev:
            raise ValueError("cal_feature_importance(pool=...): the pool must be the SortedPool of this frame's data, planned "
                             "on the current octree")
        perm = pool.perm
        coord_s, label_s, _, slots = pool.soa()  # (the sweep's entry point takes the contiguous arrays)
        coord_s, label_s = octree._check_coord(coord_s), label_s.to(torch.float32)
    else:
        perm, slots = plan_batch(octree, coord_pool, sort=True)  # sorted position j holds pool sample perm[j]
        p = perm.long()
        coord_s = octree._check_coord(coord_pool)[p].contiguous()
        label_s = label_pool[p].contiguous()
    lib = _lib.lib()
    if pool is not None:
        idx, begin, _, max_chunk = pool.importance_chunks(bs, down_rate)  # (cached: a loop may have asked for it already)
    else:
        from .sampler import importance_chunks

        idx, begin, _, max_chunk = importance_chunks(perm, sample_count, bs, down_rate)
    opts = StepOptions(sigma=float(sigma), loss_reduction=loss_reduction, decoder_grad_on=False)
    cfg = octree.step_config(sigma=float(sigma), weight_e=0.0, eikonal_on=0,
                             reduction_sum=1 if loss_reduction == "sum" else 0, decoder_grad_on=0, sorted_input=2,
                             n_global=max_chunk, kernel_variant=int(opts.kernel_variant), loss_weight_on=0,
                             inv_n=1.0)
    # the reference's post-condition (:38): the features' .grad are zero tensors (the chunks' gradients themselves live in the
    # sweep's scratch, not here)
    torch._foreach_zero_([_dense_grad(f) for f in octree.hier_features])
    rows = octree.row_counts()
    group, scratch_bytes, ws_bytes = C.c_int32(), C.c_size_t(), C.c_size_t()
    _lib.check(lib.shine_importance_sweep_sizes(len(octree.hier_features), rows, iter_n, max_chunk, _scratch_budget(dev),
                                                C.byref(group), C.byref(scratch_bytes), C.byref(ws_bytes)),
               "shine_importance_sweep_sizes")
    scratch = _zero_scratch(dev, scratch_bytes.value)
    ws = torch.empty(ws_bytes.value, dtype=torch.uint8, device=dev)
    try:
        _lib.check(
            lib.shine_importance_sweep(
                t.handle, C.byref(cfg), coord_s.data_ptr(), label_s.data_ptr(), None, idx.data_ptr(), slots.data_ptr(),
                begin, iter_n, octree.feature_ptrs(), rows,
                _lib.ptr_array([q.data_ptr() for q in mlp.fused_params()]),
                _lib.ptr_array([w.data_ptr() for w in octree.importance_weight]), group.value, scratch.data_ptr(),
                scratch.numel(), ws.data_ptr(), ws.numel(), _lib.current_stream_handle()),
            "shine_importance_sweep")
    except Exception:
        _SCRATCH.pop(dev, None)  # (a call that failed half-way may have left gradients in the scratch: it is not re-used)
        raise
