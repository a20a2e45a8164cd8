"""utils/loss.py:17-24 and utils/tools.py:175-185 under their reference names (Tier A, the strict drop-in).

`sdf_bce_loss` is ONE HIP launch that returns the loss and keeps d loss / d pred for its backward (the torch composite is a
sigmoid, a BCEWithLogits and their two backward launches); `get_gradient` on the output of the fused query_feature -> sdf node
is ONE launch of the forward kernel's closed-form d pred / d coord build, linked to that node so that the eikonal term's
backward joins the node's one fused launch (autograd_ops.InterpSdfGradCoord).  Anything else — CPU tensors, other dtypes,
other reductions, a pred that did not come from the fused node — runs the reference's torch composite, same results.
"""
import ctypes as C

import torch
import torch.nn as nn
from torch.autograd import grad

from . import _ext, _lib
from .autograd_ops import InterpSdfGradCoord


class _SdfBce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, weight, sigma, reduction_sum):
        p = pred.detach()
        l = label.detach()
        l = l if (l.dtype == torch.float32 and l.is_contiguous()) else l.contiguous().float()
        p = p if p.is_contiguous() else p.contiguous()
        w = None
        if weight is not None:
            w = weight.detach()
            w = w if (w.dtype == torch.float32 and w.is_contiguous()) else w.contiguous().float()
        n = p.shape[0]
        out = torch.empty(n + 1, dtype=torch.float32, device=p.device)  # [d loss / d pred (n) | loss]
        _lib.check(_lib.lib().shine_bce_loss(p.data_ptr(), l.data_ptr(), w.data_ptr() if w is not None else None, n, float(sigma),
                                             1 if reduction_sum else 0, out[n:].data_ptr(), out.data_ptr(),
                                             _lib.current_stream_handle()), "shine_bce_loss")
        ctx.dpred = out[:n]
        return out[n]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return ctx.dpred * g, None, None, None, None


def _bce_composite(pred, label, sigma, weight, weighted, bce_reduction):
    loss_bce = nn.BCEWithLogitsLoss(reduction=bce_reduction, weight=weight if weighted else None)
    return loss_bce(pred, torch.sigmoid(label / sigma))


def sdf_bce_loss(pred, label, sigma, weight, weighted=False, bce_reduction="mean"):
    """utils/loss.py:17-24"""
    w = weight if weighted else None
    if (pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 1 and pred.shape[0] > 0 and label.shape == pred.shape
            and label.device == pred.device and bce_reduction in ("mean", "sum") and not isinstance(sigma, torch.Tensor)
            and (w is None or (w.shape == pred.shape and w.device == pred.device)) and not label.requires_grad):
        ext = _ext.module()
        if ext is not None:  # the C++ node (csrc/shine_torch_ext.cpp)
            return ext.bce_loss(pred, label, w, float(sigma), bce_reduction == "sum")
        return _SdfBce.apply(pred, label, w, float(sigma), bce_reduction == "sum")
    return _bce_composite(pred, label, sigma, weight, weighted, bce_reduction)


def get_gradient(inputs, outputs):
    """utils/tools.py:175-185: d outputs / d inputs with create_graph=True.  For the pred of the fused query_feature -> sdf
    node and its own coord: one launch (autograd_ops.InterpSdfGradCoord); otherwise the reference's autograd call."""
    link = getattr(outputs, "_shine_link", None)
    if link is not None and link[0].coord is inputs and inputs.requires_grad and torch.is_grad_enabled():
        src, params, ext_link = link
        if ext_link is not None:  # the fused node is the C++ one: so is this (csrc/shine_torch_ext.cpp)
            ext = _ext.module()
            L = src.octree.featured_level_num
            return ext.grad_coord(src.octree._ext_state(ext), outputs, inputs, ext_link, list(params[:L]), list(params[L:]))
        return InterpSdfGradCoord.apply(outputs, inputs, src.octree, src, *params)
    d_points = torch.ones_like(outputs, requires_grad=False, device=outputs.device)
    return grad(outputs=outputs, inputs=inputs, grad_outputs=d_points, create_graph=True, retain_graph=True,
                only_inputs=True)[0]
