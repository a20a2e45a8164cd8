"""FusedAdam — the reference's optimiser (utils/tools.py:57-83) as one HIP launch per step.

setup_optimizer builds torch.optim.Adam(betas=(0.9, 0.99), eps=config.adam_eps) over
  group 0: the geo decoder's parameters, lr, weight_decay (L2)                       (:60-63)
  then one group per feature level, leaf level first, lr *= lr_level_reduce_ratio   (:68-72)
FusedAdam takes the same param_groups (so step_lr_decay, utils/tools.py:135-155, keeps working on
``opt.param_groups``) and applies the dense update to every tensor in ONE kernel, optionally clearing the grads
in the same pass.  Numerics follow torch.optim.Adam (tests compare the two on the GPU).
"""
import ctypes as C
import struct

import torch

from . import _ext, _lib
from .autograd_ops import bump_param_epoch


class FusedAdam:
    def __init__(self, param_groups, betas=(0.9, 0.99), eps=1e-15):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g["params"] = [p for p in g["params"]]
            g.setdefault("weight_decay", 0.0)
            self.param_groups.append(g)
        self.betas = betas
        self.eps = eps
        self.step_count = 0
        self.state = {}
        self._age = {}  # per-tensor step count, like torch.optim.Adam's state[p]["step"] (bias correction is per tensor)
        self._dev = None  # (step_state int64[8], lr float[n]) for graph-replayable steps
        self._dev_params = []  # the tensors those steps update (the set the device-side counter counts for)
        self._fast = None  # (params, exp_avg, exp_avg_sq, age) of the last eager step when every tensor had the same age (_step_fast)

    # ------------------------------------------------------------------ torch.optim.Optimizer's checkpoint interface
    # (utils/tools.py:200-213: save_checkpoint stores optimizer.state_dict(); shine_batch.py:232, shine_incre.py)
    _DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=False)

    def state_dict(self):
        """torch.optim.Adam's layout: {"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} with the
        parameters numbered across the groups in order — torch.optim.Adam(groups).load_state_dict() accepts it.  Steps a graph
        took on the device are counted in (read from the device-side counter; the graph-replayable state stays live)."""
        extra = {}
        if self._dev is not None:
            more = self.steps_taken() - self.step_count
            extra = {id(p): more for p in self._dev_params}
        state, groups, k = {}, [], 0
        for g in self.param_groups:
            idx = []
            for p in g["params"]:
                st = self.state.get(p)
                age = self._age.get(p, 0) + extra.get(id(p), 0)
                if st is not None and age > 0:  # (torch creates a parameter's state at its first step)
                    state[k] = {"step": torch.tensor(float(age)), "exp_avg": st[0], "exp_avg_sq": st[1]}
                idx.append(k)
                k += 1
            d = {key: val for key, val in g.items() if key != "params"}
            d.setdefault("betas", tuple(self.betas))
            d.setdefault("eps", self.eps)
            for key, val in self._DEFAULTS.items():
                d.setdefault(key, val)
            d["params"] = idx
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """The inverse (also of a torch.optim.Adam state_dict over the same groups): moments are copied to the parameters'
        devices, the groups' hyper-parameters (lr, weight_decay) are taken over."""
        saved = sd["param_groups"]
        if len(saved) != len(self.param_groups) or any(len(a["params"]) != len(b["params"])
                                                       for a, b in zip(saved, self.param_groups)):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        if self._dev is not None:
            self._fold_device_steps()
        params = {}
        for sg, g in zip(saved, self.param_groups):
            for i, p in zip(sg["params"], g["params"]):
                params[i] = p
            for key, val in sg.items():
                if key == "params":
                    continue
                if key == "betas":
                    self.betas = tuple(val)
                elif key == "eps":
                    self.eps = float(val)
                elif key in ("lr", "weight_decay") or key not in self._DEFAULTS:
                    g[key] = val
        self.state, self._age, self._fast = {}, {}, None
        for i, st in sd["state"].items():
            p = params[int(i)]
            m = st["exp_avg"].detach().to(device=p.device, dtype=p.dtype).contiguous().clone()
            v = st["exp_avg_sq"].detach().to(device=p.device, dtype=p.dtype).contiguous().clone()
            if m.shape != p.shape or v.shape != p.shape:
                raise ValueError("optimiser state of parameter %d has another shape" % int(i))
            self.state[p] = (m, v)
            self._age[p] = int(float(st["step"]))
        self.step_count = max(self._age.values(), default=0)

    def _tensors(self):
        out = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or not p.requires_grad:
                    continue  # torch's Adam skips parameters without grad (e.g. nclass_out, frozen decoder)
                st = self.state.get(p)
                if st is None:
                    st = (torch.zeros_like(p, memory_format=torch.contiguous_format),
                          torch.zeros_like(p, memory_format=torch.contiguous_format))
                    self.state[p] = st
                out.append((p, st[0], st[1], float(g["lr"]), float(g["weight_decay"])))
        return out

    def zero_grad(self, set_to_none=False):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def sync_lr(self):
        """Graph-replayable mode: push the param_groups' learning rates to the device copy the captured step reads
        (call after step_lr_decay, utils/tools.py:135-155; outside the graph)."""
        if self._dev is not None:
            ts = self._tensors()
            self._dev[1].copy_(torch.tensor([t[3] for t in ts], dtype=torch.float32), non_blocking=False)

    def steps_taken(self) -> int:
        """Optimiser steps so far (host counter, or the device counter once graph-replayable steps were used)."""
        return int(self._dev[0][0].item()) if self._dev is not None else self.step_count

    def device_state(self):
        """The device-side step state of graph-replayable steps (int64[8]; None before the first such step): hand it to the
        fused step (StepOptions.adam_state) and the step's reduction launch advances it — then call step(..., advanced=True)."""
        return None if self._dev is None else self._dev[0]

    def _make_dev_state(self, dev):
        """[0] steps taken, [1] bias corrections (floats, written by the advance), [2] / [3] beta1^t / beta2^t as doubles"""
        def bits(x):
            return struct.unpack("<q", struct.pack("<d", x))[0]

        t = self.step_count
        if t == 0:  # zeros mean "not initialised" to adam_advance: it derives the running products itself (no host copy)
            return torch.zeros(8, dtype=torch.int64, device=dev)
        return _lib.device_constants([t, 0, bits(float(self.betas[0]) ** t), bits(float(self.betas[1]) ** t), 0, 0, 0, 0],
                                     torch.int64, dev)

    def _fold_device_steps(self):
        """Fold the steps a graph took on the device back into the host counters (before the device state is dropped or
        re-made): only the tensors that were part of the device-state set took those steps — a tensor that was frozen
        meanwhile keeps its age, like torch.optim.Adam's per-parameter state["step"]."""
        taken = self.steps_taken()
        for p in self._dev_params:
            self._age[p] = self._age.get(p, 0) + (taken - self.step_count)
        self.step_count = taken
        self._dev = None
        self._dev_params = []

    @torch.no_grad()
    def step(self, zero_grad=False, graph_safe=False, advanced=False, row_flags=None):
        """`row_flags` — EXACT active-row Adam: {feature table Parameter: uint8 flags [rows + 1]} (ops.touched_flags, also
        passed to fused_train_step as `touched`, cleared when this optimiser was created).  A row whose flag is 0 has had no
        gradient since then: m = v = g = 0, which torch's Adam leaves bit for bit unchanged — it is skipped without being
        read (a 4096-point batch touches ~10^4 of a large map's 10^7 rows).  Feature groups must carry no weight decay (the
        reference's do not, utils/tools.py:68-72).
        `graph_safe`: step counter and learning rates are read from device memory (shine_adam_step_dev), so a captured
        HIP graph of this call performs step t, t+1, ... on successive replays.  Do not mix with eager steps afterwards
        without reading steps_taken().  `advanced`: the fused step of this iteration already counted the step in
        device_state() (StepOptions.adam_state): no preparation launch."""
        if not graph_safe and not row_flags and self._dev is None and self._fast is not None and self._step_fast(zero_grad):
            return
        self._fast = None
        ts = self._tensors()
        if not ts:
            return
        bump_param_epoch()  # (the parameters change behind torch's back: tensor._version does not move)
        n = len(ts)
        ages = [self._age.get(t[0], 0) for t in ts]
        if graph_safe:
            if len(set(ages)) != 1 or (self._dev is None and ages[0] != self.step_count):
                raise NotImplementedError("graph-replayable FusedAdam steps need all tensors to have the same age "
                                          "(a parameter that started receiving grads later: use eager steps)")
            dev = ts[0][0].device
            if self._dev is not None and self._dev[1].numel() != n:
                # the set of tensors that receive grads changed while the device-side counter was live (e.g. the decoder was
                # frozen / unfrozen): fold the steps the graph took back into the host counters before the state is re-made,
                # or the bias correction would silently restart from the stale host count
                self._fold_device_steps()
                ages = [self._age.get(t[0], 0) for t in ts]
                if len(set(ages)) != 1 or ages[0] != self.step_count:
                    raise NotImplementedError("graph-replayable FusedAdam steps need all tensors to have the same age "
                                              "(a parameter that started receiving grads later: use eager steps)")
            if self._dev is None:
                # int64[8]: [0] steps taken, [1] the step's bias corrections; the rest is reserved (zeros)
                self._dev = (self._make_dev_state(dev), _lib.device_constants([t[3] for t in ts], torch.float32, dev))
                self._dev_params = [t[0] for t in ts]
            for p, m, v, _, _ in ts:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise ValueError("FusedAdam needs contiguous CUDA float32 parameters and grads")
            wd = (C.c_float * n)(*[t[4] for t in ts])
            flags = self._flag_array(ts, row_flags)
            _lib.check(
                _lib.lib().shine_adam_step_dev(
                    n, _lib.ptr_array([t[0].data_ptr() for t in ts]), _lib.ptr_array([t[0].grad.data_ptr() for t in ts]),
                    _lib.ptr_array([t[1].data_ptr() for t in ts]), _lib.ptr_array([t[2].data_ptr() for t in ts]),
                    _lib.i64_array([t[0].numel() for t in ts]), self._dev[1].data_ptr(), wd, float(self.betas[0]),
                    float(self.betas[1]), float(self.eps), self._dev[0].data_ptr(),
                    (1 if zero_grad else 0) | (2 if advanced else 0), flags, _lib.current_stream_handle(),
                ),
                "shine_adam_step_dev",
            )
            return
        if self._dev is not None:  # continue the count a graph advanced on the device
            self._fold_device_steps()
            ages = [self._age.get(t[0], 0) for t in ts]
        self.step_count += 1
        if len(set(ages)) > 1:  # torch.optim.Adam semantics: one launch per distinct age (normally there is one)
            for age in sorted(set(ages)):
                self._launch([t for t, a in zip(ts, ages) if a == age], age + 1, zero_grad, row_flags)
        else:
            self._launch(ts, ages[0] + 1, zero_grad, row_flags)
        for t in ts:
            self._age[t[0]] = self._age.get(t[0], 0) + 1
        if len(set(ages)) == 1 and not row_flags:
            self._fast = ([t[0] for t in ts], [t[1] for t in ts], [t[2] for t in ts], ages[0] + 1)

    def _step_fast(self, zero_grad) -> bool:
        """The eager step of a loop in steady state (the same tensors received grads as in the last step, all of one age): the
        state lists of that step are re-used and the C++ extension validates the tensors — ~10 us of list building and checks less
        per iteration at the reference's batch size, where the host is the bound.  False: not that case (the caller goes on)."""
        ext = _ext.module()
        if ext is None:
            return False
        params, m, v, age = self._fast
        k, n, grads, lrs, wds = 0, len(params), [], [], []
        for g in self.param_groups:
            lr, wd = g["lr"], g["weight_decay"]
            for p in g["params"]:
                gr = p.grad
                if gr is None or not p.requires_grad:
                    continue
                if k >= n or params[k] is not p:
                    return False
                k += 1
                grads.append(gr)
                lrs.append(lr)
                wds.append(wd)
        if k != n:
            return False
        bump_param_epoch()
        ext.adam_step(params, grads, m, v, lrs, wds, float(self.betas[0]), float(self.betas[1]), float(self.eps), age + 1,
                      bool(zero_grad), [])
        self.step_count += 1
        ages = self._age
        for p in params:
            ages[p] = age + 1
        self._fast = (params, m, v, age + 1)
        return True

    @staticmethod
    def _flag_array(ts, row_flags):
        if not row_flags:
            return None
        by_id = {id(p): f for p, f in row_flags.items()} if isinstance(row_flags, dict) else None
        if by_id is None:
            raise ValueError("row_flags must be a dict {parameter: uint8 flag tensor}")
        ptrs = []
        for t in ts:
            f = by_id.get(id(t[0]))
            if f is not None and not (f.is_cuda and f.dtype == torch.uint8 and f.is_contiguous()
                                      and f.numel() * 8 >= t[0].numel()):
                raise ValueError("row_flags: one uint8 flag per row of the [rows, 8] table")
            ptrs.append(f.data_ptr() if f is not None else None)
        return _lib.ptr_array(ptrs)

    def _launch(self, ts, step, zero_grad, row_flags=None):
        n = len(ts)
        if n > 16:
            raise NotImplementedError("FusedAdam handles up to 16 tensors (decoder 6 + feature levels)")
        for p, m, v, _, _ in ts:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise ValueError("FusedAdam needs contiguous CUDA float32 parameters and grads")
        ext = _ext.module()
        if ext is not None:  # (one call with tensor lists instead of five ctypes pointer arrays)
            flags = []
            if row_flags:
                self._flag_array(ts, row_flags)  # (validates)
                by_id = {id(p): f for p, f in row_flags.items()}
                flags = [by_id.get(id(t[0])) for t in ts]
            ext.adam_step([t[0] for t in ts], [t[0].grad for t in ts], [t[1] for t in ts], [t[2] for t in ts], [t[3] for t in ts],
                          [t[4] for t in ts], float(self.betas[0]), float(self.betas[1]), float(self.eps), int(step),
                          bool(zero_grad), flags)
            return
        lr = (C.c_float * n)(*[t[3] for t in ts])
        wd = (C.c_float * n)(*[t[4] for t in ts])
        _lib.check(
            _lib.lib().shine_adam_step(
                n, _lib.ptr_array([t[0].data_ptr() for t in ts]), _lib.ptr_array([t[0].grad.data_ptr() for t in ts]),
                _lib.ptr_array([t[1].data_ptr() for t in ts]), _lib.ptr_array([t[2].data_ptr() for t in ts]),
                _lib.i64_array([t[0].numel() for t in ts]), lr, wd, float(self.betas[0]), float(self.betas[1]),
                float(self.eps), int(step), 1 if zero_grad else 0, self._flag_array(ts, row_flags),
                _lib.current_stream_handle(),
            ),
            "shine_adam_step",
        )


def setup_optimizer(config, octree_feat, mlp_geo_param, mlp_sem_param=None, sigma_size=None):
    """utils/tools.py:57-83 with the fused optimiser (Adam only; opt_adam is True in config defaults, :167)."""
    if getattr(config, "semantic_on", False) or getattr(config, "ray_loss", False) or not getattr(config, "opt_adam", True):
        raise NotImplementedError("fused optimiser covers the shipped configs: Adam, no semantic head, no ray loss")
    lr_cur = config.lr
    groups = []
    if mlp_geo_param is not None:
        groups.append({"params": mlp_geo_param, "lr": lr_cur, "weight_decay": config.weight_decay})
    L = config.tree_level_feat
    for i in range(L):
        groups.append({"params": [octree_feat[L - i - 1]], "lr": lr_cur})
        lr_cur *= getattr(config, "lr_level_reduce_ratio", 1.0)
    return FusedAdam(groups, betas=(0.9, 0.99), eps=getattr(config, "adam_eps", 1e-15))


def _prepare_graph_safe(self):
    """Create the device-side step state (what the first step(graph_safe=True) does) without taking a step, so that an
    iteration can be captured into a HIP graph without an eager one in front of it."""
    # the dense gradient tensors the fused step would create on its first launch and the optimiser's exp_avg / exp_avg_sq, for
    # every parameter that has neither yet: views of ONE zero-filled buffer (incremental mapping builds a new optimiser every
    # frame, shine_incre.py:107-109: 27 small fills per frame otherwise)
    fresh = [p for g in self.param_groups for p in g["params"]
             if p.requires_grad and p.grad is None and p not in self.state and p.is_cuda and p.dtype == torch.float32]
    if fresh:
        sizes = [(p.numel() + 3) // 4 * 4 for p in fresh]  # (16-byte aligned views)
        total = sum(sizes)
        flat = torch.zeros(3 * total, dtype=torch.float32, device=fresh[0].device)
        parts = flat.split(sizes * 3)  # (one call for the 3 x len(fresh) views: a slice + view each was ~120 us per frame)
        k = len(fresh)
        for i, (p, sz) in enumerate(zip(fresh, sizes)):
            n = p.numel()
            g, m, v = parts[i], parts[k + i], parts[2 * k + i]
            if n != sz:
                g, m, v = g[:n], m[:n], v[:n]
            p.grad = g.view(p.shape)
            self.state[p] = (m.view(p.shape), v.view(p.shape))
    for g in self.param_groups:
        for p in g["params"]:
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    ts = self._tensors()
    if not ts:
        raise RuntimeError("prepare_graph_safe: no parameter requires grad")
    ages = [self._age.get(t[0], 0) for t in ts]
    if len(set(ages)) != 1 or (self._dev is None and ages[0] != self.step_count):
        raise NotImplementedError("graph-replayable FusedAdam steps need all tensors to have the same age")
    if self._dev is None or self._dev[1].numel() != len(ts):
        if self._dev is not None:
            self._fold_device_steps()
        dev = ts[0][0].device
        self._dev = (self._make_dev_state(dev), _lib.device_constants([t[3] for t in ts], torch.float32, dev))
        self._dev_params = [t[0] for t in ts]
    return self._dev[0]


def _finish_iteration(self, pending, regulariser=None, next_draw=None, active_flags=None, graph=None):
    """The tail of an iteration in ONE launch (shine_finish_iteration): `pending` is the dict a
    fused_train_step(..., pending=...) filled — its partial sums are added up where they are consumed, the regulariser
    (regulariser = dict(lambda_forget, touched, out) as for ops.fused_regularization) is evaluated on the touched rows, Adam
    is applied to every tensor and the grads are cleared; next_draw = SortedPool.next_draw(...) also draws the next
    iteration's batch in the same launch.  Graph-replayable only: the step must have counted the optimiser
    step (StepOptions.adam_state = device_state()).
    active_flags (the per-level uint8 flags the step was given as `touched`): EXACT active-row Adam — the flags are kept sticky
    and rows whose flag is 0 (no gradient since this optimiser and the flags were created: m = v = g = 0, which torch's Adam
    leaves bit for bit unchanged) are not read.  The caller zeroes the flags whenever it creates the optimiser.
    graph (loop.IterationGraph): nothing is launched — the launch becomes the tail node of the library-built iteration graph."""
    if self._dev is None:
        raise RuntimeError("finish_iteration needs the device-side step state: run one step(graph_safe=True) first")
    bump_param_epoch()
    octree, decoder = pending["octree"], pending["decoder"]
    ts = self._tensors()
    by_param = {id(t[0]): (i, t) for i, t in enumerate(ts)}
    order = list(octree.hier_features) + (list(decoder.fused_params()) if pending["dec_grad"] else [])
    if len(order) != len(ts) or any(id(p) not in by_param for p in order):
        raise NotImplementedError("finish_iteration: the optimiser must hold exactly the feature tables and the decoder tensors "
                                  "that receive grads")
    if self._dev[1].numel() != len(ts):
        raise RuntimeError("finish_iteration: the set of tensors changed since the device state was made")
    sel = [by_param[id(p)] for p in order]
    for _, (p, m, v, _, _) in sel:
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
            raise ValueError("FusedAdam needs contiguous CUDA float32 parameters and grads")
    n = len(sel)
    L = len(octree.hier_features)
    cfg = pending["cfg"]
    lam, last, imp, touched, grad_on, reg_out = 0.0, None, None, None, None, None
    if regulariser is not None and float(regulariser["lambda_forget"]) != 0.0:
        lam = float(regulariser["lambda_forget"])
        keep = pending.setdefault("keep", [])  # (contiguous copies must outlive the launch)
        lasts = [t.detach().contiguous() for t in octree.features_last_frame]
        imps = [t.contiguous() for t in octree.importance_weight]
        keep += lasts + imps
        last = _lib.ptr_array([t.data_ptr() for t in lasts])
        imp = _lib.ptr_array([t.data_ptr() for t in imps])
        touched = _lib.ptr_array([t.data_ptr() for t in regulariser["touched"]])
        grad_on = (C.c_int32 * L)(*[1 if g else 0 for g in octree._reg_grad_on])
        reg_out = regulariser["out"].data_ptr()
    if active_flags is not None:
        if regulariser is not None and any(a is not b for a, b in zip(active_flags, regulariser["touched"])):
            raise ValueError("finish_iteration: active_flags must be the regulariser's touched flags")
        touched = _lib.ptr_array([t.data_ptr() for t in active_flags])
    ns = pending["n_surf"]
    lib = _lib.lib()
    entry = lib.shine_finish_iteration if graph is None else \
        (lambda *a: lib.shine_iter_graph_set_finish(graph.handle, *a[:-1]))  # (the same arguments minus the stream)
    pending.setdefault("keep", []).extend(t[k] for _, t in sel for k in (1, 2))  # (the optimiser state the launch names)
    _lib.check(
        entry(
            C.byref(cfg), pending["n"], pending["workspace"].data_ptr(), ns.data_ptr() if ns is not None else None,
            pending["loss_parts"].data_ptr(), last, imp, touched, grad_on, lam, reg_out, n,
            _lib.ptr_array([t[0].data_ptr() for _, t in sel]), _lib.ptr_array([t[0].grad.data_ptr() for _, t in sel]),
            _lib.ptr_array([t[1].data_ptr() for _, t in sel]), _lib.ptr_array([t[2].data_ptr() for _, t in sel]),
            _lib.i64_array([t[0].numel() for _, t in sel]), self._dev[1].data_ptr(), (C.c_int32 * n)(*[i for i, _ in sel]),
            (C.c_float * n)(*[t[4] for _, t in sel]), float(self.betas[0]), float(self.betas[1]), float(self.eps),
            self._dev[0].data_ptr(), C.byref(next_draw) if next_draw is not None else None,
            1 if active_flags is not None else 0, _lib.current_stream_handle()),
        "shine_finish_iteration")


FusedAdam.finish_iteration = _finish_iteration
FusedAdam.prepare_graph_safe = _prepare_graph_safe
