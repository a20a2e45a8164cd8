"""FusedAdam — the reference's optimiser (utils/tools.py:57-83) as one HIP launch per step.

setup_optimizer builds torch.optim.Adam(betas=(0.9, 0.99), eps=config.adam_eps) over
  group 0: the geo decoder's parameters, lr, weight_decay (L2)                       (:60-63)
  then one group per feature level, leaf level first, lr *= lr_level_reduce_ratio   (:68-72)
FusedAdam takes the same param_groups (so step_lr_decay, utils/tools.py:135-155, keeps working on
``opt.param_groups``) and applies the dense update to every tensor in ONE kernel, optionally clearing the grads
in the same pass.  Numerics follow torch.optim.Adam (tests compare the two on the GPU).
"""
import ctypes as C

import torch

from . import _lib


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

    @torch.no_grad()
    def step(self, zero_grad=False):
        ts = self._tensors()
        if not ts:
            return
        self.step_count += 1
        n = len(ts)
        if n > 16:
            raise NotImplementedError("FusedAdam handles up to 16 tensors (decoder 6 + feature levels)")
        for p, m, v, _, _ in ts:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise ValueError("FusedAdam needs contiguous CUDA float32 parameters and grads")
        lr = (C.c_float * n)(*[t[3] for t in ts])
        wd = (C.c_float * n)(*[t[4] for t in ts])
        _lib.check(
            _lib.lib().shine_adam_step(
                n, _lib.ptr_array([t[0].data_ptr() for t in ts]), _lib.ptr_array([t[0].grad.data_ptr() for t in ts]),
                _lib.ptr_array([t[1].data_ptr() for t in ts]), _lib.ptr_array([t[2].data_ptr() for t in ts]),
                _lib.i64_array([t[0].numel() for t in ts]), lr, wd, float(self.betas[0]), float(self.betas[1]),
                float(self.eps), self.step_count, 1 if zero_grad else 0, _lib.current_stream_handle(),
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
