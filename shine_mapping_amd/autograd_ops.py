"""The autograd nodes behind the reference's class surface.

Tier A (strict drop-in): the reference builds the interpolation out of ~30 torch ops per level and lets
autograd differentiate it, twice when the eikonal term is on (get_gradient(create_graph=True), utils/tools.py:175-185).
Here `FeatureOctree.query_feature` is OctreeInterp and `Decoder.sdf` is FusedMLP: forward, backward and
backward-of-backward are one HIP kernel each.  When the drivers' usual sequence `feature = octree.query_feature(coord);
pred = geo_mlp.sdf(feature)` (shine_batch.py:123-124) reaches `sdf` with the feature tensor untouched, the two calls collapse
into ONE node, FusedInterpSdf, whose backward is one fused launch (the Tier-B kernel fed with autograd's d loss / d pred — and,
in the eikonal configurations, with d loss / d g from the node losses.get_gradient creates) — the split nodes remain the
fallback for everything else.

Tier B: ShineTrainStep — the whole iteration (query, decode, loss, backward) as one node.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream():
    return _lib.current_stream_handle()


# Bumped whenever this package's kernels write parameters behind torch's back (FusedAdam: tensor._version does not move):
# a speculated decoder output (OctreeInterp.forward) is only used while no such write happened since.
_PARAM_EPOCH = [0]


def param_epoch():
    return _PARAM_EPOCH[0]


def bump_param_epoch():
    _PARAM_EPOCH[0] += 1


def _null_ptrs(n):
    return _lib.ptr_array([None] * n)


class OctreeInterp(torch.autograd.Function):
    """feat = sum_l interp(F_l, coord).  Inputs: coord [N,3], the octree (non-tensor), then the L feature tables."""

    @staticmethod
    def forward(ctx, coord, octree, *feats):
        from .ops import _interp_forward

        # (a training loop: the indices are computed on demand.)  The same launch also evaluates the decoder that consumed this
        # octree's features last (FeatureOctree._spec_decoder): `pred = geo_mlp.sdf(feature)` is what follows in the drivers
        # (shine_batch.py:123-124), and Decoder.sdf then has nothing left to launch (FeatureSource.speculated)
        dec = octree.__dict__.get("_spec_decoder")
        dec = dec() if dec is not None else None
        spec = None
        if dec is not None and dec.fusable and dec._params_on(coord.device):
            mlp = dec.fused_params()
            pred = torch.empty(coord.shape[0], dtype=torch.float32, device=coord.device)
            spec = (pred, dec, mlp, param_epoch(), [p._version for p in mlp])
        feat = _interp_forward(octree, coord, want_indices=False, mlp=spec[2] if spec else None,
                               pred_out=spec[0] if spec else None)
        octree.__dict__["_spec_result"] = spec
        ctx.octree = octree
        ctx.save_for_backward(coord, *feats)
        return feat

    @staticmethod
    def backward(ctx, g):
        coord, *feats = ctx.saved_tensors
        outs = OctreeInterpBackward.apply(g, coord, ctx.octree, ctx.needs_input_grad[0], *feats)
        grad_coord = outs[0] if ctx.needs_input_grad[0] else None
        grad_feats = tuple(o if need else None for o, need in zip(outs[1:], ctx.needs_input_grad[2:]))
        return (grad_coord, None) + grad_feats


class OctreeInterpBackward(torch.autograd.Function):
    """(grad_coord, dense grad_F_0..L-1) = backward of OctreeInterp; itself differentiable wrt g and the tables."""

    @staticmethod
    def forward(ctx, g, coord, octree, want_coord, *feats):
        t = octree._require_tables()
        c = octree._check_coord(coord.detach())
        g = g.detach().contiguous().float()
        n = c.shape[0]
        grad_coord = torch.zeros((n, 3), dtype=torch.float32, device=c.device)
        grad_feats = [torch.zeros_like(f, memory_format=torch.contiguous_format) for f in feats]
        cfg = octree.step_config()
        _lib.check(
            _lib.lib().shine_interp_backward(
                t.handle, C.byref(cfg), c.data_ptr(), n, _lib.ptr_array([f.data_ptr() for f in feats]),
                octree.row_counts(), g.data_ptr(), grad_coord.data_ptr() if want_coord else None,
                _lib.ptr_array([gf.data_ptr() for gf in grad_feats]), _stream(),
            ),
            "shine_interp_backward",
        )
        ctx.octree = octree
        ctx.save_for_backward(g, coord, *feats)
        return (grad_coord,) + tuple(grad_feats)

    @staticmethod
    def backward(ctx, gg_coord, *gg_feats):
        g, coord, *feats = ctx.saved_tensors
        octree = ctx.octree
        t = octree._require_tables()
        c = octree._check_coord(coord.detach())
        n = c.shape[0]
        cfg = octree.step_config()
        grad_g = None
        grad_feats = [None] * len(feats)
        if gg_coord is not None:
            gg = gg_coord.detach().contiguous().float()
            grad_g = torch.zeros((n, 8), dtype=torch.float32, device=c.device)
            dense = [torch.zeros_like(f, memory_format=torch.contiguous_format) for f in feats]
            _lib.check(
                _lib.lib().shine_interp_backward_backward(
                    t.handle, C.byref(cfg), c.data_ptr(), n, _lib.ptr_array([f.data_ptr() for f in feats]),
                    octree.row_counts(), g.data_ptr(), gg.data_ptr(), grad_g.data_ptr(),
                    _lib.ptr_array([d.data_ptr() for d in dense]), _stream(),
                ),
                "shine_interp_backward_backward",
            )
            grad_feats = dense
        if any(x is not None for x in gg_feats):
            # d(grad_F_l)/dg: grad_F_l[id] = sum w g  ->  dg += sum_l interp(gg_F_l): the forward kernel on gg tables
            tabs = [x.detach().contiguous().float() if x is not None else torch.zeros_like(f)
                    for x, f in zip(gg_feats, feats)]
            extra = torch.empty((n, 8), dtype=torch.float32, device=c.device)
            from .ops import _dummy_mlp

            _lib.check(
                _lib.lib().shine_forward(
                    t.handle, C.byref(cfg), c.data_ptr(), n, _lib.ptr_array([x.data_ptr() for x in tabs]),
                    octree.row_counts(), _lib.ptr_array([p.data_ptr() for p in _dummy_mlp(c.device)]),
                    extra.data_ptr(), None, None, None, _stream(),
                ),
                "shine_forward",
            )
            grad_g = extra if grad_g is None else grad_g + extra
        # inputs: g, coord, octree, want_coord, *feats   (no second derivative wrt coord: it is a leaf nobody reads)
        return (grad_g, None, None, None) + tuple(grad_feats)


# ----------------------------------------------------------------------------------------------------------------------
# Tier A: Decoder.sdf (model/decoder.py:49-63) as a twice-differentiable HIP op
# ----------------------------------------------------------------------------------------------------------------------


def _f32c(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


class FusedMLP(torch.autograd.Function):
    """pred[N] = w3 . relu(W2 relu(W1 feat + b1) + b2) + b3   (shine_mlp_forward).  Inputs: feat [N,8], then the six
    decoder tensors W1 [32,8], b1, W2 [32,32], b2, w3 [1,32], b3 [1].  Backward is FusedMLPBackward (itself
    differentiable), so get_gradient(create_graph=True) (utils/tools.py:175-185) works through it."""

    @staticmethod
    def forward(ctx, feat, *mlp):
        f = _f32c(feat)
        w = [_f32c(p) for p in mlp]  # alive across the launch: a temporary's block could be handed out again under it
        n = f.shape[0]
        pred = torch.empty(n, dtype=torch.float32, device=f.device)
        _lib.check(
            _lib.lib().shine_mlp_forward(f.data_ptr(), n, _lib.ptr_array([p.data_ptr() for p in w]),
                                         pred.data_ptr(), _stream()),
            "shine_mlp_forward",
        )
        ctx.save_for_backward(feat, *mlp)
        return pred

    @staticmethod
    def backward(ctx, g):
        feat, *mlp = ctx.saved_tensors
        need_w = any(ctx.needs_input_grad[1:])
        outs = FusedMLPBackward.apply(g, feat, ctx.needs_input_grad[0], need_w, *mlp)
        grad_feat = outs[0] if ctx.needs_input_grad[0] else None
        grads = tuple(o if need else None for o, need in zip(outs[1:], ctx.needs_input_grad[1:]))
        return (grad_feat,) + grads


class FusedMLPBackward(torch.autograd.Function):
    """(grad_feat [N,8], gW1, gb1, gW2, gb2, gw3, gb3) = backward of FusedMLP for g = d loss / d pred
    (shine_mlp_backward).  Differentiable wrt g and the weights through grad_feat (shine_mlp_backward_backward): the
    eikonal term's path.  Second derivatives THROUGH the weight-grad outputs are not implemented (nothing in the
    reference differentiates them) and raise."""

    @staticmethod
    def forward(ctx, g, feat, want_feat, want_w, *mlp):
        f, gc = _f32c(feat), _f32c(g)
        w = [_f32c(p) for p in mlp]
        n = f.shape[0]
        dev = f.device
        grad_feat = torch.empty((n, 8), dtype=torch.float32, device=dev) if want_feat else None
        gw = [torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for p in mlp] \
            if want_w else None
        _lib.check(
            _lib.lib().shine_mlp_backward(
                f.data_ptr(), gc.data_ptr(), n, _lib.ptr_array([p.data_ptr() for p in w]),
                grad_feat.data_ptr() if want_feat else None,
                _lib.ptr_array([t.data_ptr() for t in gw]) if want_w else None, _stream(),
            ),
            "shine_mlp_backward",
        )
        ctx.save_for_backward(g, feat, *mlp)
        ctx.set_materialize_grads(False)
        if grad_feat is None:
            grad_feat = torch.zeros((0, 8), dtype=torch.float32, device=dev)
        if gw is None:
            gw = [torch.zeros(0, dtype=torch.float32, device=dev) for _ in mlp]
        return (grad_feat,) + tuple(gw)

    @staticmethod
    def backward(ctx, gg_feat, *gg_w):
        if any(x is not None for x in gg_w):
            raise NotImplementedError("second derivatives through the decoder's weight gradients are not implemented")
        g, feat, *mlp = ctx.saved_tensors
        none = (None,) * (4 + len(mlp))
        if gg_feat is None:
            return none
        need_g = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[4:])
        if not (need_g or need_w):
            return none
        f, gc, r = _f32c(feat), _f32c(g), _f32c(gg_feat)
        w = [_f32c(p) for p in mlp]
        n = f.shape[0]
        grad_g = torch.empty(n, dtype=torch.float32, device=f.device) if need_g else None
        gw = [torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for p in mlp] \
            if need_w else None
        _lib.check(
            _lib.lib().shine_mlp_backward_backward(
                f.data_ptr(), gc.data_ptr(), r.data_ptr(), n, _lib.ptr_array([p.data_ptr() for p in w]),
                grad_g.data_ptr() if need_g else None,
                _lib.ptr_array([t.data_ptr() for t in gw]) if need_w else None, _stream(),
            ),
            "shine_mlp_backward_backward",
        )
        grads = tuple(t if need else None for t, need in zip(gw, ctx.needs_input_grad[4:])) if need_w \
            else (None,) * len(mlp)
        # inputs: g, feat, want_feat, want_w, *mlp   (relu masks are piecewise constant: nothing flows to feat)
        return (grad_g, None, None, None) + grads


# ----------------------------------------------------------------------------------------------------------------------
# Tier A, fused across the query_feature -> sdf boundary
# ----------------------------------------------------------------------------------------------------------------------


# The fused node may take a batch whose `coord` asks for a gradient (the eikonal configurations: coord.requires_grad_(True),
# shine_batch.py:119-120) only if g = d pred / d coord is then obtained through losses.get_gradient — the fused counterpart of
# utils/tools.py:175-185 that dropin installs under the reference's name.  torch.autograd.grad(create_graph=True) straight
# through the fused node still works — its backward then recomputes through the split, twice-differentiable nodes
# (_fused_split_backward) — but costs what the split nodes cost, so without that function the eikonal configurations keep the
# split nodes from the start.
FUSE_WITH_COORD_GRAD = False
# tests: the fused node's backward as ONE wave walking the batch, so that every feature-grad atomic is applied in stream order and
# an Adam trajectory through Tier A is reproducible to the bit (StepOptions.deterministic is Tier B's switch for the same launch)
DETERMINISTIC_BACKWARD = False


class FeatureSource:
    """What FeatureOctree.query_feature remembers about the tensor it returned (attached as `feature._shine_src`): enough for
    Decoder.sdf to recognise "the untouched output of query_feature" and run the fused node instead of the split ones; the
    same object links the fused node to the node get_gradient creates (InterpSdfGradCoord), whose backward leaves
    d loss / d g here for the fused launch to pick up."""

    __slots__ = ("octree", "coord", "version", "epoch", "q", "spec")

    def __init__(self, octree, coord, feature):
        self.octree, self.coord = octree, coord
        self.version = feature._version
        self.epoch = octree._tables_epoch
        self.q = None     # d loss / d (d pred / d coord), stashed by InterpSdfGradCoord.backward
        self.spec = octree.__dict__.pop("_spec_result", None)  # (pred, decoder, its parameters, parameter epoch) or None

    def speculated(self, decoder, mlp=None):
        """the decoder output query_feature's launch already computed, if it is `decoder`'s on unchanged parameters"""
        s = self.spec
        if s is None or s[1] is not decoder or s[3] != param_epoch():
            return None
        if any(a is not b or a._version != v for a, b, v in zip(mlp if mlp is not None else decoder.fused_params(), s[2], s[4])):
            return None
        return s[0]

    def fusable(self, feature) -> bool:
        return (feature._version == self.version and self.octree._tables_epoch == self.epoch
                and (FUSE_WITH_COORD_GRAD or not self.coord.requires_grad) and self.octree.featured_level_num <= 4
                and feature.dim() == 2 and feature.shape[0] == self.coord.shape[0] and feature.shape[0] > 0)


class FusedInterpSdf(torch.autograd.Function):
    """pred = Decoder.sdf(FeatureOctree.query_feature(coord)) as ONE node (shine_batch.py:123-124 with the drivers unchanged).

    forward: the decoder forward on the features query_feature already computed (shine_mlp_forward).  backward(g): the batch
    is planned (shine_plan_batch: node order + hash slots) and ONE fused launch (shine_interp_sdf_backward: the Tier-B
    kernel with d loss / d pred = g) produces every gradient — decoder backward, decoder weight grads, interpolation
    backward with one atomic per node run — into views of one freshly zeroed flat buffer, which autograd adopts as `.grad`
    (no per-table zeros_like, no index_put).  Eikonal configurations: g = get_gradient(coord, pred) is its own node
    (InterpSdfGradCoord) that takes `pred` as an input, so autograd runs ITS backward first; it leaves d loss / d g in the
    shared FeatureSource and the fused launch here backpropagates both chains at once (the eikonal build of the kernel, closed
    form of SURVEY.md §8a) — what the split nodes do in five double-backward launches.  coord itself gets no gradient (the
    reference fills coord.grad, nothing reads it)."""

    @staticmethod
    def forward(ctx, feat_values, coord, octree, src, spec_pred, *params):
        L = octree.featured_level_num
        if spec_pred is not None:  # query_feature's launch evaluated this decoder already (FeatureSource.speculated)
            pred = spec_pred
        else:
            mlp = [_f32c(p) for p in params[L:]]  # kept alive across the launch (a temporary's block could be re-used under it)
            f = _f32c(feat_values)
            n = f.shape[0]
            pred = torch.empty(n, dtype=torch.float32, device=f.device)
            _lib.check(_lib.lib().shine_mlp_forward(f.data_ptr(), n, _lib.ptr_array([p.data_ptr() for p in mlp]),
                                                    pred.data_ptr(), _stream()), "shine_mlp_forward")
        ctx.octree, ctx.src = octree, src
        ctx.save_for_backward(coord, *params)
        ctx.set_materialize_grads(False)
        return pred

    @staticmethod
    def backward(ctx, g):
        from .dp import plan_batch
        from .ops import _workspace

        if torch.is_grad_enabled():
            # a DIFFERENTIABLE backward was asked for — torch.autograd.grad(..., create_graph=True) straight through this node,
            # e.g. the reference's own get_gradient (utils/tools.py:175-185) in a driver that bound it before dropin re-bound
            # the name.  The one fused launch below is not differentiable; the split nodes are: recompute through them.
            return FusedInterpSdf._split_backward(ctx, g)
        coord, *params = ctx.saved_tensors
        octree, src = ctx.octree, ctx.src
        q, src.q = src.q, None
        L = octree.featured_level_num
        feats, mlp = params[:L], params[L:]
        need_f = [bool(x) for x in ctx.needs_input_grad[5:5 + L]]
        need_m = any(ctx.needs_input_grad[5 + L:])
        if g is None and q is None:
            return (None,) * (5 + len(params))
        t = octree._require_tables(with_ranks=True, probe=False)  # (plan_batch below is the probe, and records itself)
        c = octree._check_coord(coord.detach())
        n = c.shape[0]
        dev = c.device
        g = _f32c(g) if g is not None else torch.zeros(n, dtype=torch.float32, device=dev)
        sizes = [p.numel() if nf else 0 for p, nf in zip(feats, need_f)] + [p.numel() if need_m else 0 for p in mlp]
        flat = torch.empty((sum(sizes) + 3) // 4 * 4, dtype=torch.float32, device=dev)
        perm, slots = plan_batch(octree, c, zero=flat, sort=False)  # (the plan's first pass also clears the gradient buffer)
        views, off = [], 0
        for p, sz in zip(params, sizes):
            views.append(flat[off:off + sz].view_as(p) if sz else None)
            off += sz
        mlp_c = [_f32c(p) for p in mlp]
        feats_c = [_f32c(p) for p in feats]
        cfg = octree.step_config(sorted_input=1, decoder_grad_on=1 if need_m else 0,
                                 kernel_variant=0x4000 if DETERMINISTIC_BACKWARD else 0)
        ws = _workspace(dev, cfg)
        _lib.check(
            _lib.lib().shine_interp_sdf_backward(
                t.handle, C.byref(cfg), c.data_ptr(), perm.data_ptr(), slots.data_ptr(), g.data_ptr(),
                q.data_ptr() if q is not None else None, n,
                _lib.ptr_array([f.data_ptr() for f in feats_c]), octree.row_counts(),
                _lib.ptr_array([p.data_ptr() for p in mlp_c]),
                _lib.ptr_array([v.data_ptr() if v is not None else None for v in views[:L]]),
                _lib.ptr_array([v.data_ptr() if v is not None else None for v in views[L:]]) if need_m else None,
                ws.data_ptr(), ws.numel(), _stream(),
            ),
            "shine_interp_sdf_backward",
        )
        return (None, None, None, None, None) + tuple(views)


def _fused_split_backward(ctx, g):
    """backward of FusedInterpSdf through OctreeInterp -> FusedMLP (twice differentiable, one kernel per derivative): the
    gradients come back attached to the split nodes' graph, so whatever the caller builds on them differentiates as it would
    have without the fusion"""
    coord, *params = ctx.saved_tensors
    octree = ctx.octree
    L = octree.featured_level_num
    if g is None:
        return (None,) * (5 + len(params))
    need = [bool(ctx.needs_input_grad[1])] + [bool(x) for x in ctx.needs_input_grad[5:]]
    cand = [coord] + list(params)
    with torch.enable_grad():
        feat = OctreeInterp.apply(coord, octree, *params[:L])
        octree.__dict__.pop("_spec_result", None)  # (OctreeInterp's speculated decoder output belongs to no FeatureSource here)
        pred = FusedMLP.apply(feat, *params[L:])
        wanted = [t for t, n in zip(cand, need) if n and t.requires_grad]
        got = iter(torch.autograd.grad(pred, wanted, g, create_graph=True, allow_unused=True)) if wanted else iter(())
    grads = [next(got) if (n and t.requires_grad) else None for t, n in zip(cand, need)]
    return (None, grads[0], None, None, None) + tuple(grads[1:])


FusedInterpSdf._split_backward = staticmethod(_fused_split_backward)


class InterpSdfGradCoord(torch.autograd.Function):
    """raw = d pred / d coord [N, 3] for the pred of a FusedInterpSdf node — get_gradient(coord, pred) of utils/tools.py:175-185
    (grad_outputs = ones) without running autograd backwards: ONE launch of the forward kernel in its closed-form d pred /
    d coord build (shine_forward grad_x_out).  Differentiable: its backward hands d loss / d raw to the fused node (which
    autograd runs afterwards, because `pred` is an input here) instead of launching anything itself."""

    @staticmethod
    def forward(ctx, pred, coord, octree, src, *params):
        L = octree.featured_level_num
        t = octree._require_tables()
        c = octree._check_coord(coord.detach())
        n = c.shape[0]
        feats_c = [_f32c(p) for p in params[:L]]
        mlp_c = [_f32c(p) for p in params[L:]]
        raw = torch.empty((n, 3), dtype=torch.float32, device=c.device)
        cfg = octree.step_config(sigma=1.0)
        _lib.check(
            _lib.lib().shine_forward(t.handle, C.byref(cfg), c.data_ptr(), n, _lib.ptr_array([f.data_ptr() for f in feats_c]),
                                     octree.row_counts(), _lib.ptr_array([p.data_ptr() for p in mlp_c]), None, None, None,
                                     raw.data_ptr(), _stream()),
            "shine_forward")
        ctx.src = src
        ctx.n_in = 4 + len(params)
        ctx.set_materialize_grads(False)
        return raw

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gq):
        if gq is not None:
            q = _f32c(gq)
            ctx.src.q = q if ctx.src.q is None else ctx.src.q + q
        return (None,) * ctx.n_in


# ----------------------------------------------------------------------------------------------------------------------
# Tier A: FeatureOctree.cal_regularization as a node (the incremental configuration through the unchanged shine_incre.py)
# ----------------------------------------------------------------------------------------------------------------------


class OctreeRegularizer(torch.autograd.Function):
    """reg = sum_levels sum_{u in unique(hierarchical_indices)} importance[u] (F[u] - F_last[u])^2  — cal_regularization,
    model/feature_octree.py:246-255, called once per iteration by shine_incre.py:152-158 — for the coordinates of the octree's
    last query_feature.  The reference finds the rows with a sort-based unique() of 8 N int64 per level and runs three gathers;
    here one launch marks the rows the query's nodes address (shine_mark_touched: byte flags) and one row-parallel launch
    evaluates the sum on the flagged rows and clears the flags again (shine_regularize).
    Gradient: only the levels whose features_last_frame is still a DETACHED copy receive one (from the second frame on the
    reference stores an attached clone of the Parameter, :160, and d reg / d F cancels) — 2 importance (F - F_last) on the
    flagged rows, evaluated by the same kernel in backward.  The tensors themselves say which (levels_with_gradient)."""

    @staticmethod
    def levels_with_gradient(octree):
        """per level: True (features_last_frame is detached: the gradient is live), False (an attached clone of this level's
        Parameter: it cancels) — or None for a state this node does not cover (the copy requires grad through something else)"""
        on = []
        for last, p in zip(octree.features_last_frame, octree.hier_features):
            if not last.requires_grad:
                on.append(True)
                continue
            fn = last.grad_fn
            nf = fn.next_functions if fn is not None else ()
            if fn is None or not type(fn).__name__.startswith("CloneBackward") or len(nf) != 1 \
                    or getattr(nf[0][0], "variable", None) is not p:
                return None
            on.append(False)
        return on

    @staticmethod
    def forward(ctx, octree, coord, *feats):
        from .ops import touched_flags

        L = octree.featured_level_num
        t = octree._require_tables()
        c = octree._check_coord(coord.detach())
        dev = c.device
        rows = octree.row_counts()
        flags = octree.__dict__.get("_reg_flags")
        if flags is None or any(f.shape[0] != p.shape[0] or f.device != dev for f, p in zip(flags, feats)):
            flags = octree.__dict__["_reg_flags"] = touched_flags(octree)
            octree.__dict__["_reg_flags_dirty"] = False
        if octree.__dict__.get("_reg_flags_dirty"):  # a forward whose backward never came left its rows flagged
            for f in flags:
                f.zero_()
            octree.__dict__["_reg_flags_dirty"] = False
        lib = _lib.lib()
        cfg = octree.step_config()
        _lib.check(lib.shine_mark_touched(t.handle, C.byref(cfg), c.data_ptr(), None, None, c.shape[0], rows,
                                          _lib.ptr_array([f.data_ptr() for f in flags]), _stream()), "shine_mark_touched")
        live = OctreeRegularizer.levels_with_gradient(octree)
        need_grad = any(bool(n) and bool(on) for n, on in zip(ctx.needs_input_grad[2:], live))
        feats_c = [_f32c(p) for p in feats]
        last = [_f32c(x.detach()) for x in octree.features_last_frame]
        imp = [_f32c(x) for x in octree.importance_weight]
        out = torch.empty(1, dtype=torch.float64, device=dev)
        off = (C.c_int32 * L)(*([0] * L))
        _lib.check(
            lib.shine_regularize(L, _lib.ptr_array([x.data_ptr() for x in feats_c]), _lib.ptr_array([x.data_ptr() for x in last]),
                                 _lib.ptr_array([x.data_ptr() for x in imp]), _null_ptrs(L),
                                 _lib.ptr_array([f.data_ptr() for f in flags]), rows, off, 0.0, out.data_ptr(), 0,
                                 1 if need_grad else 0, _stream()),
            "shine_regularize")
        if need_grad:  # the flagged rows are needed again: backward clears them
            octree.__dict__["_reg_flags_dirty"] = True
            ctx.octree, ctx.flags, ctx.keep, ctx.live = octree, flags, (feats_c, last, imp), live
        ctx.need_grad = need_grad
        return out[0].to(torch.float32)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        n_in = 2 + len(ctx.needs_input_grad[2:])
        if not ctx.need_grad or g is None:
            return (None,) * n_in
        octree, flags = ctx.octree, ctx.flags
        feats_c, last, imp = ctx.keep
        L = octree.featured_level_num
        on = [bool(n) and bool(o) for n, o in zip(ctx.needs_input_grad[2:], ctx.live)]
        grads = [torch.zeros_like(p) if o else None for p, o in zip(feats_c, on)]
        scratch = torch.empty(1, dtype=torch.float64, device=feats_c[0].device)
        _lib.check(
            _lib.lib().shine_regularize(
                L, _lib.ptr_array([x.data_ptr() for x in feats_c]), _lib.ptr_array([x.data_ptr() for x in last]),
                _lib.ptr_array([x.data_ptr() for x in imp]), _lib.ptr_array([x.data_ptr() if x is not None else None for x in grads]),
                _lib.ptr_array([f.data_ptr() for f in flags]), octree.row_counts(), (C.c_int32 * L)(*[1 if o else 0 for o in on]),
                1.0, scratch.data_ptr(), 0, 0, _stream()),
            "shine_regularize")
        octree.__dict__["_reg_flags_dirty"] = False
        gf = g.to(torch.float32)
        return (None, None) + tuple(x * gf if x is not None else None for x in grads)


# ----------------------------------------------------------------------------------------------------------------------
# Tier B: the fused step as ONE autograd node
# ----------------------------------------------------------------------------------------------------------------------


class ShineTrainStep(torch.autograd.Function):
    """loss, pred, g = ShineTrainStep.apply(octree, decoder, opts, coord, sdf_label, weight, extras, *F_levels, *mlp6)

    query_feature + Decoder.sdf + sdf_bce_loss (+ get_gradient and the eikonal term) of shine_batch.py:123-185 as one
    node of the autograd graph.  forward launches the fused HIP kernel, which computes the loss AND every gradient in
    the same pass into private dense buffers; backward multiplies them by d(total)/d(loss) and hands them to autograd,
    which accumulates into `.grad` exactly as for the reference's composite (so `opt.zero_grad(); loss.backward();
    opt.step()` of shine_batch.py:208-210 is unchanged).  `pred` and `g` are returned for logging / other loss terms
    but carry no gradient (the fused node already accounts for the BCE and eikonal terms)."""

    @staticmethod
    def forward(ctx, octree, decoder, opts, coord, sdf_label, weight, extras, *params):
        from .ops import _fused_launch

        L = octree.featured_level_num
        feats, mlp = params[:L], params[L:]
        need_f = [bool(x) for x in ctx.needs_input_grad[7:7 + L]]
        need_m = any(ctx.needs_input_grad[7 + L:])
        dev = feats[0].device
        # one private flat buffer for every gradient of this node: a single fill, a single scale in backward
        sizes = [p.numel() if nf else 0 for p, nf in zip(feats, need_f)] + [p.numel() if need_m else 0 for p in mlp]
        flat = torch.zeros((sum(sizes) + 3) // 4 * 4, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p, sz in zip(params, sizes):
            views.append(flat[off:off + sz].view_as(p) if sz else None)
            off += sz
        loss, pred, g = _fused_launch(octree, decoder, coord, sdf_label, weight, opts, gfeat=views[:L],
                                      gmlp=views[L:] if need_m else [None] * 6, dec_grad=need_m, **extras)
        ctx.flat, ctx.views = flat, views
        if g is None:
            g = torch.zeros((0, 3), dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(pred, g)  # (one call: a second call would replace the first)
        return loss.to(torch.float32), pred, g

    @staticmethod
    def backward(ctx, grad_loss, _gp, _gg):
        flat, views = ctx.flat, ctx.views
        ctx.flat = ctx.views = None  # single use, like autograd's own buffers
        if flat is None:
            raise RuntimeError("ShineTrainStep: backward through the fused node a second time (its buffers are freed)")
        flat.mul_(grad_loss.to(torch.float32))
        return (None,) * 7 + tuple(views)
