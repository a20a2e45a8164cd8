"""SortedPool — LiDARDataset's sample pool kept in octree-node order, batches drawn as sorted indices (§8 f-3).

The reference keeps (coord, sdf_label, weight) pools and draws ``torch.randint`` indices every iteration
(dataset/lidar_dataset.py:430-450).  Here the pool is planned ONCE per frame (shine_plan_batch over the whole pool:
node order + every sample's hash slots), stored in that order, and ``draw(n)`` returns n sorted i.i.d. uniform indices
(shine_sample_sorted) — the same multiset distribution as randint.  ``fused_train_step(..., pool=sp, idx=idx)`` then
reads an already node-ordered batch straight out of the pool: no per-batch sort, no hashing in the fused kernel.
Re-plan (``rebuild``) whenever the octree grows: hash slots move when a table rehashes.
"""
import ctypes as C

import torch

from . import _lib
from .dp import plan_batch


def canonical_order(perm, slots):
    """Positions that put the samples of every node (= every run of identical slot rows; interchangeable misses included)
    in ascending pool-index order, runs left where they are.  perm [N] int32, slots [N, L] int32 in visiting order."""
    change = torch.ones(perm.numel(), dtype=torch.bool, device=perm.device)
    change[1:] = (slots[1:] != slots[:-1]).any(dim=1)
    key = (torch.cumsum(change, 0) << 32) + perm.long()
    return torch.argsort(key)


SURF_PARTS = 64  # include/shine_hip.h SHINE_SURF_PARTS

def importance_chunks(perm, n, bs, down_rate=1):
    """The chunks of cal_feature_importance (utils/incre_learning.py:27-31: chunk c = pool[c * bs * down_rate : (c + 1) * bs *
    down_rate : down_rate] of the ORIGINAL pool order) as segments of a node-ordered pool's sorted positions; perm[j] = original
    index of the sample at sorted position j.  -> (idx int32 [kept] on the device, begin = host int64[n_chunks + 1], n_chunks,
    largest chunk).  shine_importance_chunks: every kept sample is placed at chunk * bs + k and each chunk's segment sorted by one
    workgroup in LDS (bs <= 16384; a radix pass over the chunk ids beyond)."""
    import math

    n, bs, down_rate = int(n), int(bs), int(down_rate)
    n_chunks = math.ceil(n / (bs * down_rate))
    lib = _lib.lib()
    perm = perm.to(torch.int32).contiguous()
    idx = torch.empty(n, dtype=torch.int32, device=perm.device)
    begin = (C.c_int64 * (n_chunks + 1))()
    need = C.c_size_t()
    _lib.check(lib.shine_importance_chunks(None, n, bs, down_rate, None, None, n_chunks, None, C.byref(need), None),
               "shine_importance_chunks")
    ws = torch.empty(need.value, dtype=torch.uint8, device=perm.device)
    _lib.check(lib.shine_importance_chunks(perm.data_ptr(), n, bs, down_rate, idx.data_ptr(), begin, n_chunks, ws.data_ptr(),
                                           C.byref(need), _lib.current_stream_handle()), "shine_importance_chunks")
    largest = max((begin[c + 1] - begin[c] for c in range(n_chunks)), default=0)
    return idx, begin, n_chunks, largest


class DrawChain:
    """The whole next large sorted draw — and the next step's opt.zero_grad() — riding on every step's reduction launch
    (include/shine_hip.h shine_draw_rider): a batch-mode step is then TWO launches, the fused kernel and its reduction.

        chain = pool.draw_chain(n, idx, buckets=(flat_a, flat_b), surf=eikonal)   # idx: int32 [n], the batch every step reads
        chain.prime()                                                             # draw 0 + pass 1 of draw 1; parity 0 next
        for k in ...:
            p = chain.parity
            fused_train_step(..., opts_p (draw_rider=chain.rider[p]), n_surf=chain.surf_parts[p], pool=pool, idx=idx,
                             grad_buffers=<views of buckets[p]>)
            chain.parity ^= 1

    Step k accumulates into buckets[k & 1]; its launch clears buckets[(k + 1) & 1], draws batch k + 1 into idx and runs pass 1 of
    batch k + 2.  Capture-safe: the stream ids live in device memory; a captured graph bakes the parity of each of its steps in, so
    replay graphs only where the host's `parity` says the device is (an even number of steps per graph keeps it at 0)."""

    def __init__(self, pool, n, idx, buckets=None, surf=False):
        dev = pool.device
        n = int(n)
        if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == n):
            raise ValueError("DrawChain: idx must be a contiguous CUDA int32 tensor of n entries")
        self.pool, self.n, self.idx = pool, n, idx
        nb = (n + 1 + 1023) // 1024
        self.state = torch.zeros(2, dtype=torch.int64, device=dev)
        self.block_sum = (torch.zeros(nb, dtype=torch.float64, device=dev), torch.zeros(nb, dtype=torch.float64, device=dev))
        self.surf_parts = (pool.surf_parts_buffer(), pool.surf_parts_buffer()) if surf else (None, None)
        self.buckets = buckets
        self.parity = 0
        self.rider = []
        for p in (0, 1):
            r = _lib.DrawRider()
            r.pool_size, r.n, r.seed = pool.size, n, pool.seed
            r.state, r.parity = self.state.data_ptr(), p
            r.block_sum[0], r.block_sum[1] = self.block_sum[0].data_ptr(), self.block_sum[1].data_ptr()
            r.idx_out = idx.data_ptr()
            if surf:
                r.surf_bits = pool.surf_bits().data_ptr()
                r.surf_parts[0], r.surf_parts[1] = self.surf_parts[0].data_ptr(), self.surf_parts[1].data_ptr()
            if buckets is not None:
                z = buckets[1 - p]  # the NEXT step's bucket
                if not (z.is_cuda and z.is_contiguous() and z.data_ptr() % 16 == 0 and (z.numel() * z.element_size()) % 16 == 0):
                    raise ValueError("DrawChain: the gradient buckets must be contiguous, 16-byte aligned and sized")
                r.zero_ptr, r.zero_bytes = z.data_ptr(), z.numel() * z.element_size()
            self.rider.append(r)

    def prime(self, first_stream_id=None):
        """draw 0 into idx, pass 1 of draw 1; the caller's buckets[0] must be clean.  Stream ids: every chain of a pool gets a
        range of its own (chain number << 40, + the pool's draw count), so its batches repeat neither the stand-alone draws of
        the pool (stream ids 0, 1, 2, ...) nor another chain's; `first_stream_id` pins it (tests: the stand-alone sampler's ids)"""
        if first_stream_id is None:
            self.pool._chains = getattr(self.pool, "_chains", 0) + 1
            first_stream_id = (self.pool._chains << 40) + int(self.pool.draws)
        sid = int(first_stream_id)
        _lib.check(_lib.lib().shine_draw_rider_prime(C.byref(self.rider[0]), sid, _lib.current_stream_handle()),
                   "shine_draw_rider_prime")
        self.first_stream_id = sid
        self.parity = 0
        return self


class SortedPool:
    def __init__(self, octree, coord, sdf_label, weight, seed=42, canonical=False):
        """`canonical`: order the samples of one node by their original pool index.  The plan's counting sort places the
        samples of a node in whatever order its atomics retire, so two processes (or two runs) hold the same pool in
        different within-node orders: still a valid pool — draws stay i.i.d. uniform — but not the SAME draw.  Data-parallel
        ranks that want literally one global batch (and anyone who wants run-to-run reproducible draws) pass True; it costs
        one device sort of the pool per rebuild."""
        self.octree = octree
        self.canonical = bool(canonical)
        self.seed = int(seed)
        self.draws = 0
        self._ws = {}  # per batch size, never replaced: captured HIP graphs bake the address in
        self._stream_state = None  # device uint64[4] for graph-replayable draws (loop.GraphedIteration)
        self._rider_for = None     # size of the draw whose first pass the last fused step carried (StepOptions.next_draw)
        self.rebuild(coord, sdf_label, weight)

    def rebuild(self, coord, sdf_label, weight):
        perm, slots = plan_batch(self.octree, coord, sort=True)  # (node order whatever the pool's size: it is drawn from many times)
        if self.canonical and perm.numel() > 1:
            order = canonical_order(perm, slots)
            perm, slots = perm[order].contiguous(), slots[order].contiguous()
        p = perm.long()
        self.perm = perm  # sorted position j holds source sample perm[j] (cal_feature_importance(pool=...) re-uses the plan)
        self._surf_bits = None
        self._chunks = {}  # (bs, down_rate) -> importance_chunks(...)
        self._soa = None
        self.size = int(coord.shape[0])
        self.device = coord.device
        L = int(slots.shape[1])
        self.levels = L
        if L <= 4 and coord.is_cuda:
            # ONE 32-byte record per sample (csrc/shine_step_body.hpp RecLayout): {x, y, z, label | weight, slot[0..L-1], pad} for
            # L <= 3, {x, y, z, label | slot[0..3]} + a separate weight array for L = 4.  The fused step reads a drawn sample —
            # a sparse sorted position of a pool of 10^7-10^8 — as one cache line instead of four (coord, label, weight, slots).
            rec = torch.zeros((self.size, 8), dtype=torch.int32, device=coord.device)
            recf = rec.view(torch.float32)
            recf[:, 0:3] = coord[p]
            recf[:, 3] = sdf_label[p]
            if L < 4:
                recf[:, 4] = weight[p]
                rec[:, 5:5 + L] = slots
                self._weight_sep = None
            else:
                rec[:, 4:8] = slots
                self._weight_sep = weight[p].contiguous()
            self.rec = rec
            # the four arrays as (strided) views of the records: fine for torch indexing — pool.weight[idx.long()],
            # pool.slots[idx.long()] —; library calls that want the contiguous arrays take soa()
            self.coord, self.sdf_label = recf[:, 0:3], recf[:, 3]
            self.weight = recf[:, 4] if L < 4 else self._weight_sep
            self.slots = rec[:, 5:5 + L] if L < 4 else rec[:, 4:8]
        else:  # (more than 4 featured levels: the check library's kernel only; CPU tensors: host-side tests)
            self.rec = None
            self.coord = coord[p].contiguous()
            self.sdf_label = sdf_label[p].contiguous()
            self.weight = weight[p].contiguous()
            self.slots = slots  # already in pool (= visiting) order
        self.tables_epoch = self.octree._tables_epoch
        # sampler scratch of other pool sizes is dead weight (a graph captured for the old pool is invalid anyway: the
        # tables epoch moved); without this an object rebuilt every frame pins one buffer per distinct frame size
        for k in [k for k in self._ws if k[1] != self.size]:
            del self._ws[k]

    def soa(self):
        """(coord [P,3], sdf_label [P], weight [P], slots [P,L]) as CONTIGUOUS arrays — what the pool-mode entry points other than
        the fused step take (the importance sweep, the check library's kernel).  Materialised from the records on first use and
        kept until the next rebuild: frame pools (10^5 samples) pay nothing noticeable; a batch-mode pool of 10^8 samples should
        not need it on its hot path (the fused step, shine_mark_touched and the draw read the records)."""
        if self._soa is None:
            self._soa = tuple(t.contiguous() for t in (self.coord, self.sdf_label, self.weight, self.slots))
        return self._soa

    def importance_chunks(self, bs, down_rate=1):
        """The chunks of cal_feature_importance as segments of this pool's sorted positions (importance_chunks below), cached: it
        depends on the plan only, so an incremental loop may ask for it as soon as the pool is built and
        incre_learning.cal_feature_importance(pool=...) finds it here."""
        key = (int(bs), int(down_rate))
        hit = self._chunks.get(key)
        if hit is None:
            hit = self._chunks[key] = importance_chunks(self.perm, self.size, key[0], key[1])
        return hit

    def draw(self, n, out=None, zero=None, graph_safe=False, n_global=None, slice_begin=0, surf_parts=None,
             pass1_done=False):
        """n sorted i.i.d. uniform sample indices (int32, device).  `zero`: optional contiguous float tensor cleared in
        the same pass (the flat gradient bucket, i.e. opt.zero_grad()).  `graph_safe`: the stream id is read from (and
        advanced in) device memory, so a captured HIP graph of this call draws a fresh batch at every replay.
        Data parallel: `n_global` / `slice_begin` — this call returns draws [slice_begin, slice_begin + n) of ONE global
        sorted batch of n_global draws that every rank agrees on (same seed, same draw count): the rank's contiguous
        slice of the node-ordered global batch without generating the other ranks' indices.
        `surf_parts` (surf_parts_buffer()): the launches that write the indices also count the drawn samples with weight > 0
        (the eikonal term's surface samples, shine_batch.py:183-185) into it, as 64 partial counts — pass the tensor to
        fused_train_step as n_surf (the kernels add the parts up); in torch the count costs six launches.
        `pass1_done` (graph_safe draws of >= 16 K): the previous fused step carried this draw's first pass on its reduction
        launch (StepOptions.next_draw = pool.next_draw(...)), so the draw is ONE launch (shine_sample_sorted_finish).  Honoured
        only if that step really was the last user of the pool's device stream state (tracked on the host) — otherwise both
        passes run.  A captured graph of {draw(pass1_done=True), step with the rider} must not be interleaved with other
        graph_safe draws of the same pool between replays."""
        dev = self.device
        lib = _lib.lib()
        stream = _lib.current_stream_handle()
        sliced = n_global is not None and (int(n_global) != n or slice_begin)
        nd = int(n_global) if sliced else n  # the draw whose spacings pass 1 sums
        # the one-launch completion is valid only if the LAST thing this pool's device stream state saw was a fused step that
        # carried the first pass of exactly this draw (ops marks the pool when it launches one); otherwise both passes run
        pass1_done = bool(pass1_done and graph_safe and self._rider_for == nd)
        self._rider_for = None
        ws = self._ws.get((nd, self.size))
        none13 = (None, None)
        if ws is None:
            need = C.c_size_t(0)
            _lib.check(lib.shine_sample_sorted(self.size, nd, self.seed, 0, None, None, 0, *none13, None, C.byref(need),
                                               stream), "shine_sample_sorted")
            ws = (torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=dev), nd, int(need.value))
            self._ws[(nd, self.size)] = ws
        idx = out if out is not None else torch.empty(n, dtype=torch.int32, device=dev)
        need = C.c_size_t(ws[2])
        zero_args = (zero.data_ptr() if zero is not None else None,
                     zero.numel() * zero.element_size() if zero is not None else 0)
        surf_args = none13
        if surf_parts is not None:
            if (surf_parts.dtype != torch.int64 or not surf_parts.is_contiguous() or surf_parts.device != dev
                    or surf_parts.numel() != SURF_PARTS):
                raise ValueError("surf_parts must be a contiguous int64[%d] tensor on the pool's device "
                                 "(surf_parts_buffer())" % SURF_PARTS)
            surf_args = (self.surf_bits().data_ptr(), surf_parts.data_ptr())
        state = None
        if graph_safe:
            if self._stream_state is None:
                self._stream_state = _lib.device_constants([self.draws, 0, self.draws, 0], torch.int64, dev)
            state = self._stream_state.data_ptr()
        if pass1_done:
            _lib.check(lib.shine_sample_sorted_finish(self.size, nd, int(slice_begin) if sliced else 0, n, self.seed, state,
                                                      idx.data_ptr(), *zero_args, *surf_args, ws[0].data_ptr(), ws[2], stream),
                       "shine_sample_sorted_finish")
            return idx
        if sliced:
            _lib.check(lib.shine_sample_sorted_slice(self.size, nd, int(slice_begin), n, self.seed, self.draws, state,
                                                     idx.data_ptr(), *zero_args, *surf_args, ws[0].data_ptr(),
                                                     C.byref(need), stream), "shine_sample_sorted_slice")
        elif graph_safe:
            _lib.check(lib.shine_sample_sorted_dev(self.size, n, self.seed, state, idx.data_ptr(), *zero_args, *surf_args,
                                                   ws[0].data_ptr(), C.byref(need), stream), "shine_sample_sorted_dev")
        else:
            _lib.check(lib.shine_sample_sorted(self.size, n, self.seed, self.draws, idx.data_ptr(), *zero_args, *surf_args,
                                               ws[0].data_ptr(), C.byref(need), stream), "shine_sample_sorted")
        if not graph_safe:
            self.draws += 1
        return idx

    def draw_chain(self, n, idx, buckets=None, surf=False):
        """DrawChain(self, ...): every step's reduction launch draws the next batch (and clears the next step's bucket)"""
        return DrawChain(self, n, idx, buckets=buckets, surf=surf)

    def next_draw(self, n, out=None, surf_parts=None, n_global=None):
        """The graph-replayable draw(n, out=out, surf_parts=...) as a shine_next_draw record, for
          * FusedAdam.finish_iteration(next_draw=...): the optimiser's launch draws the NEXT iteration's batch in a few extra
            blocks (n < 16 K; `out` required);
          * StepOptions.next_draw: the fused step's reduction launch carries the first pass of the NEXT large draw (n_global:
            the size of the global draw when this rank draws a slice of it), which draw(..., pass1_done=True) completes.
        Keeps the device stream state of graph_safe draws."""
        dev = self.device
        nsp = int(n_global) if n_global is not None else int(n)
        ws = self._ws.get((nsp, self.size))
        if ws is None:
            need = C.c_size_t(0)
            _lib.check(_lib.lib().shine_sample_sorted(self.size, nsp, self.seed, 0, None, None, 0, None, None, None,
                                                      C.byref(need), _lib.current_stream_handle()), "shine_sample_sorted")
            ws = (torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=dev), nsp, int(need.value))
            self._ws[(nsp, self.size)] = ws
        if out is None:
            out = torch.empty(int(n), dtype=torch.int32, device=dev)  # (the step's rider does not write indices)
        if self._stream_state is None:
            self._stream_state = _lib.device_constants([self.draws, 0, self.draws, 0], torch.int64, dev)
        if not (out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and out.numel() == int(n)):
            raise ValueError("next_draw: out must be a contiguous CUDA int32 tensor of n entries")
        if surf_parts is not None and (surf_parts.dtype != torch.int64 or surf_parts.numel() != SURF_PARTS):
            raise ValueError("next_draw: surf_parts must be surf_parts_buffer()")
        nd = _lib.NextDraw()
        nd._pool = self  # (ops marks the pool when a step carries this record)
        nd.pool_size, nd.n, nd.seed = self.size, nsp, self.seed
        nd.workspace = ws[0].data_ptr()
        nd.stream_state, nd.idx_out = self._stream_state.data_ptr(), out.data_ptr()
        nd.surf_bits = self.surf_bits().data_ptr() if surf_parts is not None else None
        nd.surf_parts = surf_parts.data_ptr() if surf_parts is not None else None
        return nd

    def surf_bits(self):
        """one bit per pool sample: weight > 0 (a surface sample of the eikonal term) — what the draw's surface count reads
        instead of the weights themselves (1/32 of the bytes: cache resident).  Built on first use after a rebuild."""
        if self._surf_bits is None:
            pos = (self.weight > 0)
            pad = (-pos.numel()) % 32
            if pad:
                pos = torch.cat([pos, pos.new_zeros(pad)])
            w = (pos.view(-1, 32).to(torch.int64) << torch.arange(32, device=pos.device, dtype=torch.int64)).sum(dim=1)
            self._surf_bits = torch.where(w >= (1 << 31), w - (1 << 32), w).to(torch.int32).contiguous()
        return self._surf_bits

    def surf_parts_buffer(self, n=None):
        """int64[64] buffer for draw(n, surf_parts=...): the partial surface counts of a batch (SHINE_SURF_PARTS)"""
        return torch.zeros(SURF_PARTS, dtype=torch.int64, device=self.device)

    def get_batch(self, idx):
        """(coord, sdf_label, weight) of a drawn batch, for code that wants the tensors (Tier A / debugging)."""
        i = idx.long()
        if self.rec is None:
            return self.coord[i], self.sdf_label[i], self.weight[i]
        r = self.rec[i].view(torch.float32)  # (one gather of the drawn records)
        return r[:, 0:3].contiguous(), r[:, 3].contiguous(), (r[:, 4].contiguous() if self._weight_sep is None else self._weight_sep[i])
