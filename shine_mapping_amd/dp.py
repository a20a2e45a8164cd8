"""Data-parallel plumbing around the fused step and the Morton ordering of a batch.

The reference is single-GPU (utils/tools.py:26) and has no collective; this is new design:
one process per GPU, replicated feature tables + decoder, each rank runs the fused step on its shard of
the global batch, then ONE flat all-reduce(sum) of the dense grads (decoder 1377 floats + the L feature-grad
tables) over RCCL/xGMI, after which every rank applies the identical dense Adam step (replicas stay
bit-identical without a broadcast).  Normalisers use the GLOBAL batch size / surface count
(StepOptions.n_global, all_reduce_scalar) so the summed shard gradients equal the single-GPU gradient.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class GradReducer:
    """Flat-bucket gradient all-reduce.  `dist` is torch.distributed (backend nccl = RCCL on ROCm, gloo in CPU tests)."""

    def __init__(self, params, dist=None, group=None):
        self.params = list(params)
        self.dist = dist
        self.group = group
        self._flat = None
        self._views = None

    def _ensure_flat(self):
        """Re-home every .grad as a view into one flat buffer so the collective is a single large message
        (xGMI is point-to-point: few big messages beat many small ones)."""
        total = sum(p.numel() for p in self.params)
        stale = self._flat is None or self._flat.numel() != total or any(
            p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(self.params, self._views)
        )
        if not stale:
            return
        dev = self.params[0].device
        padded = torch.zeros((total + 3) // 4 * 4, dtype=torch.float32, device=dev)
        flat = padded[:total]
        self._flat_padded = padded
        views, off = [], 0
        for p in self.params:
            v = flat[off: off + p.numel()].view_as(p)
            if p.grad is not None:
                v.copy_(p.grad)
            p.grad = v
            views.append(v)
            off += p.numel()
        self._flat, self._views = flat, views

    @property
    def flat(self):
        """The flat gradient bucket (padded to a multiple of 16 bytes): pass it as plan_batch(zero=...)."""
        self._ensure_flat()
        return self._flat_padded

    def zero_grads(self):
        """One fill for every gradient (what opt.zero_grad amounts to for the fused step's dense grads)."""
        self._ensure_flat()
        self._flat.zero_()

    def all_reduce_grads(self):
        self._ensure_flat()
        if self.dist is not None:
            self.dist.all_reduce(self._flat, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce_scalar(self, t: torch.Tensor):
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class _Scratch(dict):
    """Module-level scratch registry: buffers under short keys, the byte counts the library asked for under key + (sizes...).
    Only the LAST size query per buffer is remembered (clear_sizes), so the registry does not grow with the run."""

    def clear_sizes(self, key):
        for k in [k for k in self if len(k) > len(key) and k[:len(key)] == key]:
            del self[k]


_WS = _Scratch()


def morton_order(octree, coord: torch.Tensor) -> torch.Tensor:
    """argsort of the batch by leaf-level node Morton key -> int32 perm [N] (device; no host sync)."""
    coord = octree._check_coord(coord.detach())
    n = coord.shape[0]
    cfg = octree.step_config(with_sort_box=True)
    lib = _lib.lib()
    stream = _lib.current_stream_handle()
    wide = sum(cfg.sort_bits) > 32 or min(cfg.sort_bits) <= 0
    # rocPRIM's scratch depends on the number of key bits (onesweep passes), which grows with the map's bounding box
    key = ("sort", str(coord.device))
    skey = key + (n, wide, tuple(cfg.sort_bits))
    sized = _WS.get(skey)
    if sized is None:  # size query once per (batch size, key bits); ONE grow-only buffer per device behind it
        q = C.c_size_t(0)
        _lib.check(lib.shine_morton_sort(C.byref(cfg), None, n, None, None, C.byref(q), stream),
                   "shine_morton_sort")
        _WS.clear_sizes(key)
        sized = _WS[skey] = int(q.value)
    ent = _WS.get(key)
    if ent is None or ent.numel() < sized:
        ent = _WS[key] = torch.empty(max(sized + sized // 4, 1), dtype=torch.uint8, device=coord.device)
    ws, need = ent, C.c_size_t(sized)
    perm = torch.empty(n, dtype=torch.int32, device=coord.device)
    _lib.check(
        lib.shine_morton_sort(C.byref(cfg), coord.data_ptr(), n, perm.data_ptr(), ws.data_ptr(), C.byref(need), stream),
        "shine_morton_sort",
    )
    return perm


def plan_batch(octree, coord: torch.Tensor, zero: torch.Tensor = None, _debug_variant: int = 0, out=None, sort: bool = True):
    """Order the batch by octree node (counting sort over the node ranks) and look up every point's hash slots: `perm` IS the node
    order (sort=True, the default of this public function: consumers such as sampler.canonical_order, incre_learning.
    chunk_partition or a SortedPool rely on it — ADVICE r05).  sort=False is the per-iteration form of the step's own callers
    (ops._fused_launch, the Tier A backward): batches of <= 16384 points (the reference's batch size is 4096) are left in the
    order given (perm = identity, ONE launch: the fused step does not see the order at that size and the histogram runs over every
    node of the tree); larger ones are sorted either way.

    Returns (perm [N] int32, slots [N, L] int32), both on the device, to pass to fused_train_step(perm=, slots=).
    Cheaper than morton_order (3 small launches vs a multi-pass radix sort) and it moves the hash probing out of the
    fused kernel.  `zero`: an optional contiguous float tensor cleared in the same pass (GradReducer.flat, i.e. the
    step's opt.zero_grad())."""
    t = octree._require_tables(with_ranks=True)
    coord = octree._check_coord(coord.detach())
    n = coord.shape[0]
    L = octree.featured_level_num
    cfg = octree.step_config(kernel_variant=_debug_variant | (0x800 if sort else 0))
    lib = _lib.lib()
    stream = _lib.current_stream_handle()
    # ONE grow-only scratch buffer per device.  plan_batch never runs inside a captured graph, so replacing the buffer by a
    # bigger one is safe; a cache keyed by (n, n_buckets) would pin another 20 B per sample for every distinct pool size of
    # an incremental-mapping run (every frame's pool and every octree growth is a new key).
    key = ("plan", str(coord.device))
    sized = _WS.get(key + (n, octree._n_buckets))
    if sized is None:
        q = C.c_size_t(0)
        _lib.check(lib.shine_plan_batch(t.handle, C.byref(cfg), None, n, None, None, None, 0, None, C.byref(q),
                                        stream), "shine_plan_batch")
        _WS.clear_sizes(key)
        sized = _WS[key + (n, octree._n_buckets)] = int(q.value)
    ent = _WS.get(key)
    if ent is None or ent.numel() < sized:
        ent = _WS[key] = torch.empty(max(sized + sized // 4, 1), dtype=torch.uint8, device=coord.device)
    ws, need = ent, C.c_size_t(ent.numel())
    if out is not None:  # caller-owned (e.g. double-buffered) outputs
        perm, slots = out
        assert perm.dtype == torch.int32 and perm.numel() == n and slots.dtype == torch.int32 and slots.numel() == n * L
    else:
        perm = torch.empty(n, dtype=torch.int32, device=coord.device)
        slots = torch.empty((n, L), dtype=torch.int32, device=coord.device)
    _lib.check(
        lib.shine_plan_batch(t.handle, C.byref(cfg), coord.data_ptr(), n, perm.data_ptr(), slots.data_ptr(),
                             zero.data_ptr() if zero is not None else None,
                             zero.numel() * zero.element_size() if zero is not None else 0,
                             ws.data_ptr(), C.byref(need), stream),
        "shine_plan_batch",
    )
    octree._tables_read_done()  # (asynchronous growth, FeatureOctree.enable_async_growth: the plan is a probe of the tables)
    return perm, slots


def mark_touched(octree, pool, idx: torch.Tensor, flags=None):
    """Byte flags (one uint8 tensor [rows_s + 1] per level, top-down) of the feature rows the pool samples `idx` address:
    shine_mark_touched on the node-ordered pool's slot table.  Under data parallelism every rank calls this on the
    GLOBAL draw (same seed everywhere), so all ranks hold the same row set without a collective."""
    t = octree._require_tables(probe=False)  # (memoised slots)
    if flags is None:
        flags = [torch.zeros(p.shape[0], dtype=torch.uint8, device=p.device) for p in octree.hier_features]
    if getattr(pool, "rec", None) is not None:  # a pool of 32-byte records: the slots are read out of the records
        cfg, base, slots = octree.step_config(sorted_input=3), pool.rec, pool.rec
    else:
        cfg, base, slots = octree.step_config(sorted_input=2), pool.coord, pool.slots
    _lib.check(
        _lib.lib().shine_mark_touched(t.handle, C.byref(cfg), base.data_ptr(), idx.data_ptr(), slots.data_ptr(),
                                      idx.numel(), octree.row_counts(), _lib.ptr_array([f.data_ptr() for f in flags]),
                                      _lib.current_stream_handle()),
        "shine_mark_touched",
    )
    return flags


class TouchedRowReducer(GradReducer):
    """Gradient exchange of the rows a step touched ("shared-feature grads") instead of the dense tables.

    The dense bucket is Σ_l rows_l·F·4 bytes whatever the batch (13 MB for a 100 m street, 0.4 GB for a 6 km map); a
    step only writes the rows its points' corners address.  Given per-level byte flags of the rows the GLOBAL batch
    touches — identical on every rank: each rank marks the global draw itself (mark_touched), or the flags were OR-reduced
    — every rank packs those rows (+ the L trash rows + the 1377 decoder floats) in the same order into one contiguous
    message, all-reduces it, and unpacks.  Rows outside the set hold zeros on every rank and stay untouched.  Packing is
    index plumbing around the collective (torch index_select / index_copy_, one host read of the row count per step), so
    the same code runs over gloo on CPU tensors in the tests.  `feature_params`: the L feature tables (top-down);
    `other_params`: the decoder tensors."""

    def __init__(self, feature_params, other_params, dist=None, group=None):
        super().__init__(list(feature_params) + list(other_params), dist, group)
        self.n_feat = len(list(feature_params))
        self.last_rows = 0      # rows exchanged by the last sparse reduce (all levels)
        self.last_bytes = 0     # message size of the last reduce

    def dense_bytes(self):
        return sum(p.numel() for p in self.params) * 4

    def or_reduce_flags(self, flags):
        """For callers whose ranks drew independent batches: make the per-rank flags the global union (MAX of bytes)."""
        if self.dist is not None:
            for f in flags:
                self.dist.all_reduce(f, op=self.dist.ReduceOp.MAX, group=self.group)
        return flags

    def _device_exchange(self, flags):
        """CUDA path: row lists, pack and unpack are library kernels (shine_touched_index / _pack / _unpack): ~10 launches
        and ONE host read (the L row counts = the message size) around the collective."""
        feats = self.params[:self.n_feat]
        L = self.n_feat
        dev = feats[0].device
        lib = _lib.lib()
        stream = _lib.current_stream_handle()
        rows = [int(p.shape[0]) - 1 for p in feats]
        st = getattr(self, "_dev_state", None)
        if st is None or st["rows"] != rows or st["dev"] != dev:
            need = C.c_size_t(0)
            _lib.check(lib.shine_touched_index(L, None, _lib.i64_array(rows), None, None, None, C.byref(need), stream),
                       "shine_touched_index")
            st = dict(rows=rows, dev=dev, ws=torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=dev),
                      ws_bytes=int(need.value), idx=[torch.empty(max(r, 1), dtype=torch.int32, device=dev) for r in rows],
                      counts=torch.zeros(L, dtype=torch.int64, device=dev), rows_arr=_lib.i64_array(rows))
            self._dev_state = st
        need = C.c_size_t(st["ws_bytes"])
        flag_ptrs = _lib.ptr_array([f.data_ptr() for f in flags])
        idx_ptrs = _lib.ptr_array([t.data_ptr() for t in st["idx"]])
        _lib.check(lib.shine_touched_index(L, flag_ptrs, st["rows_arr"], idx_ptrs, st["counts"].data_ptr(),
                                           st["ws"].data_ptr(), C.byref(need), stream), "shine_touched_index")
        counts = st["counts"].cpu().tolist()  # the message size: the one host read of the exchange
        n_rows = int(sum(counts))
        F = feats[0].shape[1]
        n_dec = sum(p.numel() for p in self.params[self.n_feat:])
        feat_total = sum(p.numel() for p in feats)
        msg = torch.empty((n_rows + L) * F + n_dec, dtype=torch.float32, device=dev)
        grad_ptrs = _lib.ptr_array([p.grad.data_ptr() for p in feats])
        counts_arr = _lib.i64_array(counts)
        _lib.check(lib.shine_touched_pack(L, grad_ptrs, idx_ptrs, counts_arr, st["rows_arr"], msg.data_ptr(), stream),
                   "shine_touched_pack")
        if n_dec:  # the decoder grads sit behind the feature tables in the flat bucket: one contiguous copy
            msg[(n_rows + L) * F:].copy_(self._flat[feat_total:feat_total + n_dec])
        self.last_rows, self.last_bytes = n_rows, msg.numel() * 4
        if self.dist is not None:
            self.dist.all_reduce(msg, op=self.dist.ReduceOp.SUM, group=self.group)
        _lib.check(lib.shine_touched_unpack(L, grad_ptrs, idx_ptrs, counts_arr, st["rows_arr"], flag_ptrs, msg.data_ptr(),
                                            stream), "shine_touched_unpack")
        if n_dec:
            self._flat[feat_total:feat_total + n_dec].copy_(msg[(n_rows + L) * F:])

    def all_reduce_touched(self, flags):
        """flags: per level uint8 [rows_l + 1] (or [rows_l]); identical on all ranks.  Clears them for the next step."""
        self._ensure_flat()
        feats = self.params[:self.n_feat]
        if feats[0].is_cuda:
            return self._device_exchange(flags)
        idx = []
        for p, f in zip(feats, flags):
            f = f[: p.shape[0] - 1]
            idx.append(torch.nonzero(f, as_tuple=False).flatten())  # (host read of the count: the message size)
        pieces = [p.grad.index_select(0, i).reshape(-1) for p, i in zip(feats, idx)]
        pieces += [p.grad[-1].reshape(-1) for p in feats]                      # trash rows: every miss lands there
        pieces += [p.grad.reshape(-1) for p in self.params[self.n_feat:]]      # decoder
        msg = torch.cat(pieces)
        self.last_rows = int(sum(i.numel() for i in idx))
        self.last_bytes = msg.numel() * 4
        if self.dist is not None:
            self.dist.all_reduce(msg, op=self.dist.ReduceOp.SUM, group=self.group)
        off = 0
        F = feats[0].shape[1]
        for p, i in zip(feats, idx):
            k = i.numel() * F
            p.grad.index_copy_(0, i, msg[off:off + k].view(-1, F))
            off += k
        for p in feats:
            p.grad[-1].copy_(msg[off:off + F])
            off += F
        for p in self.params[self.n_feat:]:
            p.grad.copy_(msg[off:off + p.numel()].view_as(p))
            off += p.numel()
        for f in flags:
            f.zero_()


class RowGatherReducer(GradReducer):
    """Gradient exchange by all-gather of the rows EACH RANK touched (the reference is single-GPU).

    Under the sorted global draw a rank's slice is a contiguous stretch of the node-ordered batch: it touches ~1/world of the
    feature rows, and the ranks' row sets overlap only at the slice borders.  Instead of all-reducing the dense bucket (or
    the union of all ranks' rows, TouchedRowReducer), every rank MOVES the rows its own step flagged (ids + values, fixed
    capacity) plus the decoder grads into one message, the ranks all-gather the messages, and every rank adds all of them
    back into its bucket: ONE collective per exchange, about half the bytes of an all-reduce, no flag OR-reduce, no host
    read — the whole step {draw, fused step, exchange} is capturable in a HIP graph.

        red = RowGatherReducer(octree.hier_features, decoder.fused_params(), dist)
        fused_train_step(..., touched=red.flags)      # the step marks the rows it touches
        red.exchange()                                 # grads now hold the global sums

    exchange() may be called after every MICRO-batch of a step (the pack moves the rows out of the bucket, so the next
    micro-batch accumulates from zero and nothing is counted twice); finish() then adds everything back — with
    async_op=True the all-gather of micro-batch k runs under the fused kernel of micro-batch k + 1.  `capacity_rows`:
    rows per message (default: measured on the first exchange, x 1.5); a rank that touches more sets the overflow flag,
    which `overflowed()` reads (one host sync — call it outside the hot loop; the result of such a step is incomplete).
    CPU tensors (gloo tests) run the same algorithm in torch index ops."""

    def __init__(self, feature_params, other_params, dist=None, group=None, capacity_rows=None, async_op=False,
                 synthetic_world=0):
        super().__init__(list(feature_params) + list(other_params), dist, group)
        # measurement aid (bench.py's kitti-dp8-rank leg): without a process group, stand in for `synthetic_world` ranks — the
        # all-gather becomes that many device copies of this rank's own message, and every copy is unpacked and added: the
        # pack, the receive-side writes and the unpack of an N-rank exchange are all there, only the wire is not
        self.synthetic_world = int(synthetic_world) if dist is None else 0
        self.peer_messages = None  # synthetic world: int32 [(world - 1) * words], the other ranks' packed messages (else: copies)
        self.async_op = bool(async_op)  # issue the all-gather asynchronously (it then runs on the backend's own stream, under
        #                                 whatever the caller launches next — the next micro-batch); finish() waits for it
        self.n_feat = len(list(feature_params))
        feats = self.params[:self.n_feat]
        self.F = int(feats[0].shape[1])
        self.level_rows = [int(p.shape[0]) for p in feats]            # rows_l + 1 (the trash row is the last one)
        self.n_rows = sum(self.level_rows)
        self.tail_n = sum(p.numel() for p in self.params[self.n_feat:])
        dev = feats[0].device
        self._flags_flat = torch.zeros(self.n_rows, dtype=torch.uint8, device=dev)
        self.flags, off, keep = [], 0, []
        for r in self.level_rows:  # per-level views for fused_train_step(touched=...) / shine_mark_touched
            self.flags.append(self._flags_flat[off: off + r])
            keep.append(off + r - 1)
            off += r
        self._keep = keep
        self._flags_flat[torch.tensor(keep, device=dev)] = 1  # every miss lands in a trash row: always exchanged
        self.capacity = None if capacity_rows is None else self._round_cap(capacity_rows)
        self._pending = []       # gathered messages not yet added back (micro-batches)
        self._overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ws = None
        self.last_rows = 0
        self.last_bytes = 0

    def _round_cap(self, rows):
        return max(4, min((int(rows) + 3) // 4 * 4, (self.n_rows + 3) // 4 * 4))

    def dense_bytes(self):
        return sum(p.numel() for p in self.params) * 4

    def world(self):
        if self.synthetic_world > 1:
            return self.synthetic_world
        return self.dist.get_world_size(self.group) if self.dist is not None else 1

    def _measure_capacity(self):
        n = int(self._flags_flat.count_nonzero())  # (one host read, once)
        if self.dist is not None:  # every rank must use the same message size
            t = torch.tensor([n], dtype=torch.int64, device=self._flags_flat.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            n = int(t)
        self.capacity = self._round_cap(int(1.5 * n) + 1024)

    def _words(self):
        return 4 + self.capacity + self.capacity * self.F + self.tail_n

    def pack(self):
        """MOVE this rank's flagged rows (ids + values; the bucket's copies are cleared), the trash rows and the decoder grads
        into one fixed-size message -> int32 [4 + capacity + capacity * F + tail_n] (word 0: rows held, word 1: overflow)."""
        self._ensure_flat()
        if self.capacity is None:
            self._measure_capacity()
        flat, dev = self._flat_padded, self._flat_padded.device
        tail_off = self.n_rows * self.F
        msg = torch.empty(self._words(), dtype=torch.int32, device=dev)
        if flat.is_cuda:
            lib = _lib.lib()
            stream = _lib.current_stream_handle()
            if self._ws is None:
                need = C.c_size_t(0)
                _lib.check(lib.shine_rows_pack(None, self.n_rows, None, 0, None, 0, self.tail_n, self.capacity, None, None,
                                               C.byref(need), stream), "shine_rows_pack")
                self._ws = (torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=dev), int(need.value))
            need = C.c_size_t(self._ws[1])
            _lib.check(lib.shine_rows_pack(self._flags_flat.data_ptr(), self.n_rows, _lib.i64_array(self._keep), len(self._keep),
                                           flat.data_ptr(), tail_off, self.tail_n, self.capacity, msg.data_ptr(),
                                           self._ws[0].data_ptr(), C.byref(need), stream), "shine_rows_pack")
        else:
            self._pack_torch(flat, msg, tail_off)
        self.last_bytes = self._words() * 4
        return msg

    def add_messages(self, gathered, world, first=True):
        """add `world` messages (one all-gather's result, rank order) into the bucket; first=False: a later micro-batch of the
        same step (its decoder tail is added to, not written over, what the first one left)"""
        self._ensure_flat()
        flat = self._flat_padded
        tail_off = self.n_rows * self.F
        if flat.is_cuda:
            # tail_n is part of the message stride for EVERY message; the later micro-batches of a step add their tails
            _lib.check(_lib.lib().shine_rows_unpack_add(gathered.data_ptr(), world, self.capacity, flat.data_ptr(), tail_off,
                                                        self.tail_n, 0 if first else 1, self._overflow.data_ptr(),
                                                        _lib.current_stream_handle()), "shine_rows_unpack_add")
        else:
            self._unpack_torch(flat, gathered, world, tail_off, first)

    def exchange(self, finish=True):
        """pack this rank's flagged rows (moving them out of the bucket) and all-gather; finish=True also adds every
        gathered message back (finish=False after all but the last micro-batch of a step)"""
        msg = self.pack()
        dev = msg.device
        world = self.world()
        words = self._words()
        work = None
        if self.dist is not None:  # (also with one rank: the collective is then a copy, and the same code path is exercised)
            gathered = torch.empty(world * words, dtype=torch.int32, device=dev)
            if self.async_op:
                work = self.dist.all_gather_into_tensor(gathered, msg, group=self.group, async_op=True)
            else:
                self.dist.all_gather_into_tensor(gathered, msg, group=self.group)
        elif self.synthetic_world > 1:
            if self.peer_messages is not None:  # the other ranks' REAL messages of one step (benchlib.run_dp_rank)
                gathered = torch.cat([msg, self.peer_messages])
            else:
                gathered = msg.repeat(self.synthetic_world)
        else:
            gathered = msg
        self._pending.append((gathered, work, msg))  # (msg is kept alive until the collective has read it)
        if finish:
            self.finish()

    def finish(self):
        """add every gathered message (all ranks, all micro-batches) back into the bucket"""
        world = self.world()
        first = True
        for gathered, work, _ in self._pending:
            if work is not None:
                work.wait()  # the current stream waits for the collective (no host block on the NCCL / RCCL backend)
            self.add_messages(gathered, world, first)
            first = False
        self._pending = []

    def overflowed(self):
        """True if any rank of any exchange since the last call touched more rows than a message holds (host sync)"""
        v = bool(int(self._overflow.item()))
        self._overflow.zero_()
        return v

    # ---- the same algorithm in torch index ops (CPU tensors: the gloo tests)
    def _pack_torch(self, flat, msg, tail_off):
        F, cap = self.F, self.capacity
        ids = torch.nonzero(self._flags_flat, as_tuple=False).flatten()
        total = int(ids.numel())
        self.last_rows = total
        ids = ids[:cap]
        n = int(ids.numel())
        msg.zero_()
        msg[0], msg[1] = n, 1 if total > cap else 0
        msg[4: 4 + n] = ids.to(torch.int32)
        table = flat[: self.n_rows * F].view(self.n_rows, F)
        vals = msg[4 + cap: 4 + cap + cap * F].view(torch.float32).view(cap, F)
        vals[:n] = table[ids]
        table[ids] = 0
        msg[4 + cap + cap * F:].view(torch.float32).copy_(flat[tail_off: tail_off + self.tail_n])
        flat[tail_off: tail_off + self.tail_n] = 0
        self._flags_flat.zero_()
        self._flags_flat[torch.tensor(self._keep)] = 1

    def _unpack_torch(self, flat, gathered, world, tail_off, first):
        F, cap, words = self.F, self.capacity, self._words()
        table = flat[: self.n_rows * F].view(self.n_rows, F)
        tail = torch.zeros(self.tail_n, dtype=torch.float32)
        for r in range(world):
            m = gathered[r * words: (r + 1) * words]
            n = int(m[0])
            if int(m[1]):
                self._overflow.fill_(1)
            ids = m[4: 4 + n].long()
            vals = m[4 + cap: 4 + cap + cap * F].view(torch.float32).view(cap, F)[:n]
            table.index_add_(0, ids, vals)
            tail += m[4 + cap + cap * F:].view(torch.float32)
        if first:
            flat[tail_off: tail_off + self.tail_n] = tail
        else:
            flat[tail_off: tail_off + self.tail_n] += tail
