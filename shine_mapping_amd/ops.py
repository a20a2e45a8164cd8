"""Operators of the SHINE hot path on top of libshine_hip.so.

Tier B (fused, benchmarked):  ``fused_train_step`` = query + decode + sdf_bce_loss (+ eikonal) + the
whole backward in one HIP pass, writing dense ``.grad`` tensors exactly where autograd would
(shine_batch.py:115-209 minus the optimiser).

Tier A (strict drop-in):      ``octree_interp`` = FeatureOctree.query_feature as an autograd op
(model/feature_octree.py:237-244) so the reference drivers run unchanged.

Every function here calls the HIP library; none has a CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _ext, _lib, autograd_ops


@dataclass
class StepOptions:
    """Scalars of one training iteration (names follow utils/config.py)."""

    sigma: float                      # sigma_sigmoid = logistic_gaussian_ratio*sigma_sigmoid_m*scale (shine_batch.py:87)
    loss_reduction: str = "mean"      # "mean" | "sum" (shine_incre.py:77-78)
    ekional_loss_on: bool = False     # (sic) config key of the reference
    weight_e: float = 0.1
    loss_weight_on: bool = False      # BCEWithLogitsLoss(weight=|weight|), utils/loss.py:18-19 (False in all shipped yamls)
    n_global: Optional[int] = None    # global batch size under data parallelism (defaults to local N)
    decoder_grad_on: Optional[bool] = None  # default: any decoder parameter requires grad (freeze_model, tools.py:188)
    deterministic: bool = False       # tests: ONE wave walks the whole batch, so the fp32 atomics of the feature-grad scatter are
                                      # applied in stream order and two runs agree to the bit (hundreds of times slower)
    # iteration hooks (loop.GraphedIteration): housekeeping of the calls that follow the step rides on its reduction launch
    adam_state: Optional[torch.Tensor] = None   # FusedAdam's device step state: the step counts the optimiser step
    adam_betas: tuple = (0.9, 0.99)
    zero_f64: Optional[torch.Tensor] = None     # one float64 element cleared by the step (the regulariser's accumulator)
    next_draw: Optional[object] = None          # SortedPool.next_draw(...): the first pass of the NEXT large sorted draw rides on
                                                # the step's reduction launch; complete it with pool.draw(..., pass1_done=True)
    draw_rider: Optional[object] = None         # sampler.DrawChain.rider[parity]: the WHOLE next sorted draw and the next step's
                                                # zero-fill ride on this step's reduction launch (cfg.draw_rider)
    kernel_variant: int = 0           # 0 the fused step (5 / 6: its far / near build whatever the table size); tests / tools: 1 the
                                      # lane-per-point reference kernel
                                      # (libshine_check.so)


def eik_needs_count(opts) -> bool:
    return bool(opts.ekional_loss_on)


def _stream():
    return _lib.current_stream_handle()


def _f32(t, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError("%s must be a CUDA float32 tensor" % name)
    return t.contiguous()


def _dense_grad(p: torch.Tensor) -> torch.Tensor:
    """autograd hands the optimiser a dense zero-initialised grad of the parameter's shape; so do we."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def forward_sdf(octree, decoder, coord, want_feat=False, want_indices=False, want_grad_x=False, sigma=1.0):
    """query_feature + Decoder.sdf, forward only (utils/mesher.py:69-72 style inference).

    Returns dict(pred[N], feat[N,8]|None, indices (bottom-up list)|None, grad_x[N,3]|None = d pred/d coord * sigma).
    """
    t = octree._require_tables()
    coord = octree._check_coord(coord.detach())
    n = coord.shape[0]
    dev = coord.device
    pred = torch.empty(n, dtype=torch.float32, device=dev)
    feat = torch.empty((n, 8), dtype=torch.float32, device=dev) if want_feat else None
    gx = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_grad_x else None
    idx = None
    if want_indices:
        idx = [torch.empty((n, 8), dtype=torch.int64, device=dev) for _ in range(octree.featured_level_num)]
    cfg = octree.step_config(sigma=float(sigma))  # (set_zero: the forward kernel re-zeroes the trash rows)
    mlp = [p.detach() for p in decoder.fused_params()]
    _lib.check(
        _lib.lib().shine_forward(
            t.handle, C.byref(cfg), coord.data_ptr(), n, octree.feature_ptrs(), octree.row_counts(),
            _lib.ptr_array([p.data_ptr() for p in mlp]),
            feat.data_ptr() if feat is not None else None, pred.data_ptr(),
            _lib.ptr_array([o.data_ptr() for o in idx]) if idx is not None else None,
            gx.data_ptr() if gx is not None else None, _stream(),
        ),
        "shine_forward",
    )
    octree._tables_read_done()
    if idx is not None:
        octree.hierarchical_indices = idx
    return dict(pred=pred, feat=feat, indices=idx, grad_x=gx)


def fused_train_step(octree, decoder, coord, sdf_label, weight, opts: StepOptions, want_grad_x=False, perm=None,
                     n_surf: Optional[torch.Tensor] = None, slots: Optional[torch.Tensor] = None, touched=None,
                     pool=None, idx: Optional[torch.Tensor] = None, pending: Optional[dict] = None, graph=None,
                     grad_buffers=None):
    """One training iteration's forward+backward (no optimiser): the fused Tier-B step, raw form.

    coord [N,3], sdf_label [N], weight [N] (sign: + surface / - free space, utils/data_sampler.py:102-103).
    Accumulates into ``.grad`` of octree.hier_features[*] and the six decoder tensors (dense, trash row
    included — what ``cur_loss.backward()`` produces, shine_batch.py:209).  Returns (loss, pred, g) where
    loss is a 0-dim float64 device tensor (no host sync, NO grad_fn — the gradients are already in place) and
    g = get_gradient(coord,pred)*sigma or None.  `train_step` is the same launch as an autograd node.

    `pending` (a dict): the step launches its fused kernel only (cfg->defer_reduce) and leaves the per-workgroup partial sums
    — decoder grads, trash-row grads, loss terms — in the workspace for the optimiser's launch; the dict is filled with what
    FusedAdam.finish_iteration(pending, ...) needs, and `loss` is valid after that call.
    `graph` (loop.IterationGraph, with `pending`): nothing is launched — the launch this call would make becomes the step node
    of the library-built iteration graph (shine_iter_graph_set_step); outputs are valid after the graph ran.
    `grad_buffers` = (L feature-grad tensors, 6 decoder-grad tensors): accumulate into THESE (zero-filled by the caller) instead of
    the parameters' `.grad`.
    """
    gfeat, gmlp = grad_buffers if grad_buffers is not None else (None, None)
    return _fused_launch(octree, decoder, coord, sdf_label, weight, opts, want_grad_x=want_grad_x, perm=perm,
                         n_surf=n_surf, slots=slots, touched=touched, pool=pool, idx=idx, pending=pending, graph=graph,
                         gfeat=gfeat, gmlp=gmlp)


def train_step(octree, decoder, coord, sdf_label, weight, opts: StepOptions, want_grad_x=False, perm=None,
               n_surf: Optional[torch.Tensor] = None, slots: Optional[torch.Tensor] = None, touched=None,
               pool=None, idx: Optional[torch.Tensor] = None):
    """The fused step as ONE node of the autograd graph (autograd_ops.ShineTrainStep; SURVEY.md §8b Tier B):

        loss, pred, g = train_step(octree, geo_mlp, coord, sdf_label, weight, opts)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()          # shine_batch.py:208-210, unchanged

    `loss` (0-dim float32, requires grad) = sdf_bce_loss [+ weight_e * eikonal]; other terms (e.g. lambda *
    cal_regularization, shine_incre.py:156-158) can be added to it before backward().  Same keyword arguments as
    fused_train_step.  When nothing requires grad (or under torch.no_grad) it degenerates to a forward pass."""
    from .autograd_ops import ShineTrainStep

    params = list(octree.feature_list()) + list(decoder.fused_params())
    extras = dict(want_grad_x=want_grad_x, perm=perm, n_surf=n_surf, slots=slots, touched=touched, pool=pool, idx=idx)
    loss, pred, g = ShineTrainStep.apply(octree, decoder, opts, coord, sdf_label, weight, extras, *params)
    return loss, pred, (g if g.numel() else None)


def _fused_launch(octree, decoder, coord, sdf_label, weight, opts: StepOptions, want_grad_x=False, perm=None,
                  n_surf: Optional[torch.Tensor] = None, slots: Optional[torch.Tensor] = None, touched=None,
                  pool=None, idx: Optional[torch.Tensor] = None, gfeat=None, gmlp=None, dec_grad=None, pending=None,
                  graph=None):
    """shine_train_step on explicit gradient buffers (gfeat: L tensors or None entries, gmlp: 6 tensors; default: the
    parameters' own dense `.grad`)."""
    # (the fused kernel reads the tables through the batch's memoised hash slots; only the check library's v0 probes in-kernel)
    t = octree._require_tables(probe=(int(opts.kernel_variant) & 0xff) == 1)
    pool_mode = pool is not None
    if pool_mode:  # batch = pool[idx] with idx sorted (sampler.SortedPool.draw): read straight out of the pool
        if idx is None or not (idx.is_cuda and idx.dtype == torch.int32):
            raise ValueError("pool mode needs idx = SortedPool.draw(n) (CUDA int32)")
        if pool.tables_epoch != octree._tables_epoch:
            raise RuntimeError("the octree grew since the pool was planned: call SortedPool.rebuild()")
        if eik_needs_count(opts) and n_surf is None:
            n_surf = (pool.weight[idx.long()] > 0).sum()
    variant = int(opts.kernel_variant) & 0xff
    # a pool of 32-byte records (sampler.SortedPool.rec; cfg.sorted_input 3): the record base goes in as `coord` — and, unread, as
    # `slots`: the records carry them —, the weight array only for 4-level trees; the check library's kernel takes the arrays
    rec_mode = pool_mode and getattr(pool, "rec", None) is not None and variant != 1
    if rec_mode:
        coord, sdf_label, weight, perm, slots = pool.rec, None, pool._weight_sep, idx, pool.rec
    elif pool_mode:
        coord, sdf_label, weight, slots = pool.soa()
        perm = idx
    if not rec_mode:
        coord = octree._check_coord(coord.detach())
    n = idx.numel() if pool_mode else coord.shape[0]
    dev = coord.device
    if not pool_mode and slots is None and variant != 1:
        # a batch without a plan (what the reference's get_batch hands over): neighbouring lanes would hit unrelated nodes and
        # the fused kernel takes the hash slots from the plan — order it by octree node and look the slots up first
        # (shine_plan_batch: 3 small launches; measured at 2^18 points x 4 levels 320 -> 143 us for the whole call against
        # in-kernel probing of the unordered batch, tools/unordered_bench.py).  A caller's own `perm` without slots (e.g.
        # dp.morton_order) is replaced by the plan's node order: the step does not depend on the visiting order.
        from .dp import plan_batch

        perm, slots = plan_batch(octree, coord, sort=False)  # (the step does not see the order of a small batch: one launch)
    if not rec_mode:
        sdf_label = _f32(sdf_label, "sdf_label")
    eik = bool(opts.ekional_loss_on)
    if bool(opts.loss_weight_on) and weight is None and not rec_mode:
        raise ValueError("loss_weight_on needs the sample weights")
    if (eik and not rec_mode) or weight is not None:
        weight = _f32(weight, "weight")
    if opts.loss_reduction not in ("mean", "sum"):
        raise ValueError("loss_reduction must be 'mean' or 'sum'")
    n_global = int(opts.n_global) if opts.n_global else n
    params = decoder.fused_params()
    if dec_grad is None:
        dec_grad = opts.decoder_grad_on
    if dec_grad is None:
        dec_grad = any(p.requires_grad for p in params)
    cfg = octree.step_config(
        sigma=float(opts.sigma), weight_e=float(opts.weight_e), eikonal_on=1 if eik else 0,
        reduction_sum=1 if opts.loss_reduction == "sum" else 0, decoder_grad_on=1 if dec_grad else 0,
        sorted_input=(3 if rec_mode else 2) if pool_mode else (0 if perm is None else 1), n_global=n_global,
        kernel_variant=int(opts.kernel_variant) | (0x4000 if getattr(opts, "deterministic", False) else 0),
        loss_weight_on=1 if opts.loss_weight_on else 0,
        inv_n=(1.0 if opts.loss_reduction == "sum" else 1.0 / max(n_global, 1)),
    )
    if opts.adam_state is not None:
        cfg.adam_state = opts.adam_state.data_ptr()
        cfg.adam_beta1, cfg.adam_beta2 = float(opts.adam_betas[0]), float(opts.adam_betas[1])
    if opts.zero_f64 is not None:
        cfg.zero_f64 = opts.zero_f64.data_ptr()
    if opts.next_draw is not None:
        cfg.next_draw = C.pointer(opts.next_draw)
    if opts.draw_rider is not None:
        if pending is not None or graph is not None or opts.next_draw is not None:
            raise ValueError("draw_rider is a hook of the plain fused step (not with next_draw / the deferred-reduction forms)")
        cfg.draw_rider = C.pointer(opts.draw_rider)
    if eik and n_surf is None:
        n_surf = (weight > 0).sum()  # stays on the device; under DP the caller all-reduces it first
    if eik and n_surf.numel() > 1:  # the sampler's per-block partial counts (SortedPool.draw(surf_parts=...))
        if n_surf.dtype != torch.int64 or not n_surf.is_contiguous():
            raise ValueError("n_surf parts must be a contiguous int64 tensor")
        if (int(opts.kernel_variant) & 0xff) == 1 or slots is None:
            n_surf = n_surf.sum()  # the check library's kernels take one count
        else:
            cfg.n_surf_parts = int(n_surf.numel())
    pred = torch.empty(n, dtype=torch.float32, device=dev)
    gx = torch.empty((n, 3), dtype=torch.float32, device=dev) if (want_grad_x and eik) else None
    loss_parts = torch.empty(4, dtype=torch.float64, device=dev)  # overwritten by the step; set_zero is in-kernel
    if gfeat is None:
        gfeat = [_dense_grad(p) if p.requires_grad else None for p in octree.feature_list()]
    if gmlp is None:
        gmlp = [_dense_grad(p) for p in params] if dec_grad else [None] * 6
    if perm is not None and not (perm.is_cuda and perm.dtype == torch.int32 and perm.numel() == n):
        raise ValueError("perm must be a CUDA int32 tensor of N entries")
    # (slots without perm: a batch that already IS in visiting order — e.g. gathered from a node-ordered pool — with its hash slots)
    if slots is not None and not pool_mode and not (slots.is_cuda and slots.dtype == torch.int32
                                                    and slots.numel() == n * octree.featured_level_num):
        raise ValueError("slots (from dp.plan_batch, or of a batch in visiting order): CUDA int32 [N, L]")
    ws = _workspace(dev, cfg)
    if pending is not None:
        if variant not in (0, 4, 5, 6) or slots is None:
            raise ValueError("pending= (deferred reduction) needs the product kernel on a planned / pool batch")
        cfg.defer_reduce = 1
        pending.clear()
        pending.update(cfg=cfg, n=n, workspace=ws, n_surf=n_surf, loss_parts=loss_parts, dec_grad=bool(dec_grad),
                       octree=octree, decoder=decoder, pred=pred, gx=gx)
    if graph is not None and pending is None:
        raise ValueError("graph= records the deferred-reduction step: pass pending= too")
    # the product library trains on <= 4 featured levels (every shipped yaml: 3 or 4) with kernel_variant 0 / 4; the
    # lane-per-point reference kernel (1: any batch, <= 8 levels) is the check library's — tests / tools ask for it by name
    if variant != 1 and octree.featured_level_num > 4:
        raise NotImplementedError("the fused step handles tree_level_feat <= 4 (every shipped config); deeper trees only run "
                                  "on the check library's reference kernel (StepOptions.kernel_variant = 1, tests / tools)")
    library = _lib.check_lib() if variant == 1 else _lib.lib()
    entry = library.shine_train_step if graph is None else \
        (lambda *a: library.shine_iter_graph_set_step(graph.handle, *a[:-1]))  # (the same arguments minus the stream)
    _lib.check(
        entry(
            t.handle, C.byref(cfg), coord.data_ptr(), sdf_label.data_ptr() if sdf_label is not None else None,
            weight.data_ptr() if weight is not None else None,
            perm.data_ptr() if perm is not None else None,
            slots.data_ptr() if slots is not None else None,
            n_surf.data_ptr() if n_surf is not None else None, n,
            octree.feature_ptrs(), octree.row_counts(),
            _lib.ptr_array([p.data_ptr() for p in params]),
            pred.data_ptr(), gx.data_ptr() if gx is not None else None,
            _lib.ptr_array([g.data_ptr() if g is not None else None for g in gfeat]),
            _lib.ptr_array([g.data_ptr() if g is not None else None for g in gmlp]),
            loss_parts.data_ptr(),
            _lib.ptr_array([x.data_ptr() for x in touched]) if touched is not None else None,
            ws.data_ptr(), ws.numel(), _stream(),
        ),
        "shine_train_step", library,
    )
    if opts.next_draw is not None and pending is None and getattr(opts.next_draw, "_pool", None) is not None:
        opts.next_draw._pool._rider_for = int(opts.next_draw.n)  # its reduction launch carried that draw's first pass
    return loss_parts[3], pred, gx  # 0-dim float64 view: BCE (+ weight_e * eikonal), no extra launch


def octree_interp(octree, coord):
    """FeatureOctree.query_feature (model/feature_octree.py:237-244) -> [N,8]; also fills hierarchical_indices."""
    needs_grad = torch.is_grad_enabled() and (
        coord.requires_grad or any(p.requires_grad for p in octree.hier_features)
    )
    if needs_grad:
        ext = _ext.module()
        if ext is not None and coord.is_cuda and octree.featured_level_num <= 4:
            return _ext_query(ext, octree, coord)  # the C++ node (csrc/shine_torch_ext.cpp)
        return autograd_ops.OctreeInterp.apply(coord, octree, *octree.feature_list())
    return _interp_forward(octree, coord)


def _ext_query(ext, octree, coord):
    """autograd_ops.OctreeInterp.forward on the C++ side: shine_forward, with the decoder that consumed this octree's features
    last riding on the launch (FeatureSource.speculated)."""
    octree._require_tables()
    dec = octree.__dict__.get("_spec_decoder")
    dec = dec() if dec is not None else None
    mlp = dec.fused_params() if (dec is not None and dec.fusable) else []
    if mlp and not dec._params_on(coord.device, mlp):
        mlp = []
    st = octree._ext_state(ext)
    octree._reg_rider(st)  # (incremental mapping: cal_regularization's value rides on this launch — set up once per frame)
    feat, pred, reg = ext.query_feature(st, coord, octree.feature_list(), mlp)
    octree.__dict__["_spec_result"] = (pred, dec, mlp, autograd_ops.param_epoch(), [p._version for p in mlp]) if mlp else None
    octree._defer_indices(coord)
    octree.__dict__["_reg_riding"] = reg  # None, or this query's regulariser (a view of the rider's ring: clone to keep)
    return feat


def _interp_forward(octree, coord, want_indices=True, mlp=None, pred_out=None):
    """query_feature's forward (shine_forward: interpolation only).  want_indices=False: hierarchical_indices is not written
    (the octree computes it from `coord` if somebody reads it, FeatureOctree.hierarchical_indices)."""
    t = octree._require_tables()
    c = octree._check_coord(coord.detach())
    n = c.shape[0]
    feat = torch.empty((n, 8), dtype=torch.float32, device=c.device)
    idx = [torch.empty((n, 8), dtype=torch.int64, device=c.device) for _ in range(octree.featured_level_num)] \
        if want_indices else None
    cfg = octree.step_config()
    _lib.check(
        _lib.lib().shine_forward(
            t.handle, C.byref(cfg), c.data_ptr(), n, octree.feature_ptrs(), octree.row_counts(),
            _lib.ptr_array([p.data_ptr() for p in mlp]) if mlp is not None else None, feat.data_ptr(),
            pred_out.data_ptr() if pred_out is not None else None,
            _lib.ptr_array([o.data_ptr() for o in idx]) if idx is not None else None, None, _stream(),
        ),
        "shine_forward",
    )
    if idx is not None:
        octree.hierarchical_indices = idx
    else:
        octree._defer_indices(c)
    return feat


_DUMMY = {}
_WORKSPACE = {}


def _workspace(dev, cfg):
    """Scratch for the fused step (per-workgroup partial sums): one buffer per (device, stream), for the life of the process.

    It is allocated ONCE at the size of the largest launch geometry (the workgroup count is capped, so the bound is ~3 MB)
    and never replaced: a captured HIP graph (loop.GraphedIteration, bench.py) bakes its address in, so swapping it for a
    bigger one later would leave the graph writing into freed memory.  Per STREAM, because the partial sums of a step live
    in it between the step's two launches: two steps in flight on different streams (a graph replaying on its capture
    stream next to eager launches, two models in one process) must not share them."""
    key = (str(dev), int(_stream() or 0))
    ws = _WORKSPACE.get(key)
    if ws is None:
        nbytes = int(_lib.lib().shine_train_step_workspace_bytes(C.byref(cfg), -1))  # bound for any batch size
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _WORKSPACE[key] = ws
    return ws


def _dummy_mlp(dev):
    key = str(dev)
    if key not in _DUMMY:
        _DUMMY[key] = [torch.zeros(s, dtype=torch.float32, device=dev) for s in (256, 32, 1024, 32, 32, 1)]
    return _DUMMY[key]


def fused_regularization(octree, lambda_forget: float, touched, out=None, out_zeroed=False, keep_flags=False):
    """FeatureOctree.cal_regularization (model/feature_octree.py:246-255) on the rows the last fused step touched.

    Returns the UNWEIGHTED regulariser as a 0-dim float64 device tensor and adds lambda * d(reg)/dF into the feature
    grads of the levels whose features_last_frame copy is detached (octree._reg_grad_on; the reference's attached
    clone, :160, contributes value only).  `touched`: the per-level uint8 flag tensors passed to fused_train_step.
    `keep_flags`: the flags double as the optimiser's sticky active-row flags (FusedAdam.step(row_flags=touched)) — they are
    left as they are, and only the rows flagged by THIS iteration's step (bit 0) enter the regulariser."""
    import ctypes as _C

    L = octree.featured_level_num
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=octree.feature_list()[0].device)
        out_zeroed = False
    feats = octree.feature_list()
    last = [t.detach().contiguous() for t in octree.features_last_frame]
    imp = [t.contiguous() for t in octree.importance_weight]
    grads = [_dense_grad(p) for p in feats]
    grad_on = (_C.c_int32 * L)(*[1 if g else 0 for g in octree._reg_grad_on])
    _lib.check(
        _lib.lib().shine_regularize(
            L, _lib.ptr_array([t.data_ptr() for t in feats]), _lib.ptr_array([t.data_ptr() for t in last]),
            _lib.ptr_array([t.data_ptr() for t in imp]), _lib.ptr_array([g.data_ptr() for g in grads]),
            _lib.ptr_array([t.data_ptr() for t in touched]), octree.row_counts(), grad_on, float(lambda_forget),
            out.data_ptr(), 1 if out_zeroed else 0, 1 if keep_flags else 0, _stream(),
        ),
        "shine_regularize",
    )
    return out[0]


def touched_flags(octree):
    """Per-level uint8 row flags for fused_train_step(touched=...) / fused_regularization (zero-initialised; views of one
    buffer: one fill)."""
    feats = octree.feature_list()
    sizes = [(int(p.shape[0]) + 15) // 16 * 16 for p in feats]
    flat = torch.zeros(sum(sizes), dtype=torch.uint8, device=feats[0].device)
    out, off = [], 0
    for p, sz in zip(feats, sizes):
        out.append(flat[off:off + int(p.shape[0])])
        off += sz
    return out
