"""GraphedIteration — one training iteration captured in a HIP graph and replayed (launch-bound inner loops belong in
hipGraphs).

The reference's inner loop (shine_batch.py:105-210, shine_incre.py:114-181) at its own batch size (4096) is bound by the
host: ~7 kernel launches per iteration at ~100 us of Python/driver time against ~75 us of GPU time.  This helper captures

    loss = fused_train_step(..., pool=pool, idx=idx)          query + decode + loss + backward (the fused kernel only)
    opt.finish_iteration(...)                                 sums of the kernel's partial vectors + [regulariser,
                                                              incremental mapping only] + fused dense Adam + grads cleared
                                                              + idx = pool.draw(n) for the NEXT iteration (sorted batch
                                                              from the node-ordered pool; the first one is drawn up front)

once and replays it.  The scalars that change per iteration — the sampler's stream id and Adam's step count — live in
device memory and are advanced by the kernels themselves (shine_sample_sorted_dev / shine_adam_step_dev), so every
replay draws a fresh batch and applies the right bias correction.  Learning-rate decay stays outside the graph
(`opt.sync_lr()` after changing `param_groups`).  Re-create after `octree.update()` (parameters are re-allocated, like
the optimiser itself, shine_incre.py:108-109).
"""
import ctypes as C

import torch

from . import _lib
from .autograd_ops import bump_param_epoch
from .ops import StepOptions, fused_regularization, fused_train_step, touched_flags


_CAPTURE_STREAM = {}


def _capture(fn):
    """Capture fn() into a new HIP graph on a side stream -> (graph, fn's result).  What `with torch.cuda.graph(g)` does,
    minus its gc.collect() + torch.cuda.empty_cache(): incremental mapping re-captures the iteration every frame
    (the parameters are re-allocated when the octree grows), and emptying the allocator's cache each time makes the next
    frame's allocations go back to hipMalloc."""
    dev = torch.cuda.current_device()
    side = _CAPTURE_STREAM.get(dev)
    if side is None:
        side = _CAPTURE_STREAM[dev] = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize(dev)  # nothing of the eager warm-up (non-blocking copies, allocator activity) straddles the capture
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    out, err = None, None
    try:
        with torch.cuda.stream(side):
            g.capture_begin()
            try:
                out = fn()
            except BaseException as e:  # keep the ORIGINAL error: ending a broken capture may raise one of its own
                err = e
            try:
                g.capture_end()
            except BaseException as e2:
                if err is None:
                    err = e2
    finally:
        main.wait_stream(side)  # the main stream re-joins the side stream on every path
    if err is not None:
        raise err
    return g, out


_WARMED = set()  # devices on which an iteration has run eagerly in this process


class IterationGraph:
    """A library-built HIP graph of `unroll` iterations (include/shine_hip.h shine_iter_graph_*): kernel nodes created from the
    launches fused_train_step(pending=..., graph=self) and FusedAdam.finish_iteration(..., graph=self) would make, re-bound in
    place (commit) when the buffers change — a new frame of incremental mapping costs no capture and no instantiation."""

    _cache = {}

    def __init__(self, unroll):
        self.unroll = int(unroll)
        self.handle = C.c_void_p()
        self.owner = None  # the GraphedIteration whose buffers the nodes currently name
        _lib.check(_lib.lib().shine_iter_graph_create(self.unroll, C.byref(self.handle)), "shine_iter_graph_create")
        self.image = None  # the decoder's operand image for small-batch steps: torch's allocation, handed to the graph

    def ensure_image(self, dev):
        if self.image is None or self.image.device != torch.device(dev):
            lib = _lib.lib()
            self.image = torch.empty(int(lib.shine_iter_graph_operand_image_floats()), dtype=torch.float32, device=dev)
            _lib.check(lib.shine_iter_graph_set_operand_image(self.handle, self.image.data_ptr()), "shine_iter_graph_set_operand_image")

    @classmethod
    def shared(cls, device, unroll, slot=0):
        """one graph per (device, unroll, slot) for the life of the process: incremental mapping builds a GraphedIteration per
        frame.  `slot`: a loop that lets the host run a frame ahead of the device alternates two slots — re-binding a graph waits
        for that graph's own last replay (shine_iter_graph_commit), and the other slot's replays are the ones still running"""
        key = (str(device), int(unroll), int(slot))
        g = cls._cache.get(key)
        if g is None:
            g = cls._cache[key] = cls(unroll)
        return g

    def commit(self):
        _lib.check(_lib.lib().shine_iter_graph_commit(self.handle), "shine_iter_graph_commit")

    def launch(self, replays):
        _lib.check(_lib.lib().shine_iter_graph_launch(self.handle, int(replays), _lib.current_stream_handle()),
                   "shine_iter_graph_launch")

    def stats(self):
        c, b = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().shine_iter_graph_stats(self.handle, C.byref(c), C.byref(b)), "shine_iter_graph_stats")
        return int(c.value), int(b.value)

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().shine_iter_graph_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


class GraphedIteration:
    """`eager_first` (default True): the constructor runs iteration 1 eagerly (it allocates the optimiser state and loads the
    kernels outside the capture), so a frame of K iterations is the constructor + run(K - 1).  False: nothing runs in the
    constructor — the optimiser's device state is created explicitly and the graph is captured straight away, so a frame is
    run(K); honoured only once an eager iteration has run on the device in this process (incremental mapping re-creates this
    object every frame: the eager iteration costs ~0.1 ms of launches that a replay does in a third of the time)."""

    def __init__(self, octree, decoder, pool, opt, opts: StepOptions, n: int, lambda_forget: float = 0.0, unroll: int = 1,
                 fold: bool = True, eager_first: bool = True, active_rows: bool = True, native: bool = True, graph_slot: int = 0):
        self.octree, self.decoder, self.pool, self.opt, self.opts, self.n = octree, decoder, pool, opt, opts, int(n)
        self.lambda_forget = float(lambda_forget)
        self.graph_slot = int(graph_slot)
        self.fold = bool(fold)  # the iteration's tail as one launch (False: reduction, regulariser and Adam as three)
        self.regularize = self.lambda_forget != 0.0
        can_fold = self.fold and hasattr(opt, "finish_iteration") and hasattr(opt, "prepare_graph_safe")
        # EXACT active-row Adam (FusedAdam.finish_iteration(active_flags=...)): rows that have had no gradient since `opt` was
        # created are not read.  Valid for an optimiser that is new (this object creates the flags cleared, so it must be built
        # together with the optimiser, as shine_incre.py:107-109 does every frame) and whose feature groups carry no weight decay
        # (the reference's groups, utils/tools.py:68-72).
        feat_ids = {id(p) for p in octree.hier_features}
        feat_wd = [float(g.get("weight_decay", 0.0)) for g in getattr(opt, "param_groups", [])
                   if any(id(p) in feat_ids for p in g["params"])]
        fresh = getattr(opt, "steps_taken", lambda: 1)() == 0
        self.active_rows = bool(active_rows and can_fold and fresh and feat_wd and all(w == 0.0 for w in feat_wd))
        self.touched = touched_flags(octree) if (self.regularize or self.active_rows) else None
        self._epoch = octree._tables_epoch
        self._idx = torch.empty(self.n, dtype=torch.int32, device=pool.coord.device)
        # the eikonal term's surface count: per-block partial counts written by the draw itself, summed by the step's kernels
        self._surf = pool.surf_parts_buffer(self.n) if opts.ekional_loss_on else None
        self.loss = self.reg = None
        self._reg_out = torch.zeros(1, dtype=torch.float64, device=pool.coord.device) if self.regularize else None
        self._hooked = None  # StepOptions with the iteration hooks (made once the optimiser has its device state)
        self._ahead = False
        dev_key = str(pool.coord.device)
        # `native`: the graph is BUILT by the library from the two launches of an iteration and re-bound in place for this
        # object's buffers (IterationGraph) instead of being captured from the stream — nothing to warm up, nothing to capture:
        # the constructor runs no iteration whatever `eager_first` says.  Needs the two-launch iteration (fold, the next draw
        # inside the tail: n < 16 K).
        self.native = bool(native and can_fold and self.n + 1 <= 16 * 1024)
        self.ran_eager = (not self.native) and (bool(eager_first) or dev_key not in _WARMED or not can_fold)
        if can_fold:
            # the optimiser's device-side step state up front: the eager warm-up iteration then runs the SAME two launches the
            # graph replays (with active rows it has to: its rows must end up flagged "touched earlier")
            self.opt.prepare_graph_safe()
        if self.ran_eager:
            # eager warm-up: allocates workspaces and optimiser state and loads the kernels outside the capture
            self.loss, self.reg = self._body()
            _WARMED.add(dev_key)
        # From here on the optimiser's launch also draws the NEXT iteration's batch (a few extra blocks, no launch of its own):
        # an iteration is {fused kernel, tail}.  `_idx` then holds the batch of the iteration to come; it is primed here.
        self._ahead = (self.fold and hasattr(self.opt, "finish_iteration") and hasattr(self.opt, "device_state")
                       and self.opt.device_state() is not None and self.n + 1 <= 16 * 1024)
        if self._ahead:
            self.pool.draw(self.n, out=self._idx, graph_safe=True, surf_parts=self._surf)
        # `unroll` iterations in ONE graph: an iteration is two small launches, and every graph boundary leaves the GPU idle
        # for ~8 us — run(n) replays the long graph n // unroll times.  A captured node costs ~16 us, so it pays when the
        # boundaries saved outweigh the nodes captured (a frame of 50 iterations: unroll 5, bench.py --unroll).  Both graphs
        # are captured when first needed: a frame whose length is a multiple of `unroll` never captures the one-iteration graph.
        self.unroll = max(1, int(unroll))
        # `loss` / `reg` always name the outputs of the launch that ran LAST (eager, one-iteration graph or unrolled graph:
        # each capture has its own output tensors)
        self.graph = self.graph_k = None
        self._out_1 = self._out_k = None
        self._native = {}  # unroll -> (IterationGraph, outputs, keep-alive) bound to this object's buffers

    def _graph_1(self):
        if self.graph is None:
            self.graph, self._out_1 = _capture(self._body)
        return self.graph

    def _graph_unrolled(self):
        if self.graph_k is None:
            def body_k():
                out = None
                for _ in range(self.unroll):
                    out = self._body()
                return out

            self.graph_k, self._out_k = _capture(body_k)
        return self.graph_k

    def _bound(self, unroll):
        """the library-built graph of `unroll` iterations, its kernel nodes naming this object's buffers"""
        ent = self._native.get(unroll)
        g = ent[0] if ent is not None else IterationGraph.shared(self.pool.coord.device, unroll, self.graph_slot)
        if ent is None or g.owner is not self:
            g.ensure_image(self.pool.coord.device)
            keep = {}
            out = self._body(graph=g, keep=keep)
            g.commit()
            g.owner = self
            ent = self._native[unroll] = (g, out, keep)
        return ent

    def _body(self, graph=None, keep=None):
        idx = self._idx if self._ahead else self.pool.draw(self.n, out=self._idx, graph_safe=True, surf_parts=self._surf)
        n_surf = self._surf
        # Iteration hooks: the step's reduction launch also counts the optimiser step (+ bias corrections) and clears the
        # regulariser's accumulator, so neither costs a launch of its own (an iteration at N = 4096 is a chain of small
        # launches: each one removed saves its run time and ~2 us of dependency gap).  The optimiser's device state exists
        # after its first graph-safe step — the eager warm-up call below runs unhooked and creates it.
        state = self.opt.device_state() if hasattr(self.opt, "device_state") else None
        hooked = state is not None
        if hooked and (self._hooked is None or self._hooked.adam_state is not state):
            import copy

            self._hooked = copy.copy(self.opts)
            self._hooked.adam_state, self._hooked.adam_betas = state, tuple(self.opt.betas)
            self._hooked.zero_f64 = self._reg_out
        opts = self._hooked if hooked else self.opts
        # hooked (every iteration but the eager warm-up): the fused kernel only, then ONE launch for {sums of its partial
        # vectors, regulariser, Adam, clearing the grads} (FusedAdam.finish_iteration) — 3 launches per iteration instead of 5
        fold = hooked and self.fold and hasattr(self.opt, "finish_iteration")
        pending = {} if fold else None
        if graph is not None and not (fold and self._ahead):
            raise RuntimeError("the library-built iteration graph holds the two-launch iteration (fold, next draw in the tail)")
        loss, _, _ = fused_train_step(self.octree, self.decoder, None, None, None, opts, n_surf=n_surf, pool=self.pool,
                                      idx=idx, touched=self.touched, pending=pending, graph=graph)
        if keep is not None:
            keep["pending"] = pending  # (every device buffer the nodes name stays alive with this object)
        if fold:
            self.opt.finish_iteration(pending, dict(lambda_forget=self.lambda_forget, touched=self.touched,
                                                    out=self._reg_out) if self.regularize else None,
                                      next_draw=self.pool.next_draw(self.n, self._idx, self._surf) if self._ahead else None,
                                      active_flags=self.touched if self.active_rows else None, graph=graph)
            return loss, (self._reg_out[0] if self.regularize else None)
        reg = None
        if self.regularize:
            reg = fused_regularization(self.octree, self.lambda_forget, self.touched, out=self._reg_out, out_zeroed=hooked)
        self.opt.step(zero_grad=True, graph_safe=True, advanced=hooked)
        return loss, reg

    def __call__(self):
        """Run one iteration; returns the loss of the fused terms as a 0-dim device tensor (no host sync)."""
        if self.octree._tables_epoch != self._epoch:
            raise RuntimeError("the octree grew since this iteration was captured: build a new GraphedIteration")
        bump_param_epoch()
        if self.native:
            g, out, _ = self._bound(1)
            g.launch(1)
            self.loss, self.reg = out
            return self.loss
        self._graph_1().replay()
        self.loss, self.reg = self._out_1
        return self.loss

    def run(self, n_iters: int):
        """n_iters iterations: replays of the `unroll`-iteration graph, the remainder one by one.  Returns the last loss."""
        if self.octree._tables_epoch != self._epoch:
            raise RuntimeError("the octree grew since this iteration was captured: build a new GraphedIteration")
        bump_param_epoch()  # (replays update the parameters without torch noticing)
        k = self.unroll if self.unroll > 1 else 0
        if self.native:
            if k and n_iters >= k:
                g, out, _ = self._bound(k)
                g.launch(n_iters // k)
                self.loss, self.reg = out
                n_iters %= k
            if n_iters:
                g, out, _ = self._bound(1)
                g.launch(n_iters)
                self.loss, self.reg = out
            return self.loss
        while k and n_iters >= k:
            self._graph_unrolled().replay()
            self.loss, self.reg = self._out_k
            n_iters -= k
        for _ in range(n_iters):
            self._graph_1().replay()
            self.loss, self.reg = self._out_1
        return self.loss
