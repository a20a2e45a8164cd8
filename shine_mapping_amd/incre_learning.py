"""cal_feature_importance — utils/incre_learning.py:8-40 on the fused HIP path.

Per chunk of the frame's pool the reference runs query + decode + BCE + backward and then
``importance_weight[i] += hier_features[i].grad.abs(); grad.zero_(); importance_weight[i][-1] *= 0``.
Here the chunk's forward+backward is one fused step with the decoder frozen (only feature grads are needed) and the
epilogue is one kernel per level (shine_importance_accumulate).
"""
import math

import torch

from . import _lib
from .dp import plan_batch
from .ops import StepOptions, _dense_grad, fused_train_step


def cal_feature_importance(data, octree, mlp, sigma, bs, down_rate=1, loss_reduction="mean", loss_weight_on=False):
    """Same signature as the reference; `data` needs .coord_pool and .sdf_label_pool (utils/incre_learning.py:14-26)."""
    # loss_weight_on has no effect here, as in the reference: it passes weight=None (utils/incre_learning.py:32), and
    # nn.BCEWithLogitsLoss(weight=None) is the unweighted loss
    sample_count = data.coord_pool.shape[0]
    batch_interval = bs * down_rate
    iter_n = math.ceil(sample_count / batch_interval)
    opts = StepOptions(sigma=float(sigma), loss_reduction=loss_reduction, decoder_grad_on=False)
    stream = _lib.current_stream_handle()
    for p in octree.hier_features:
        if p.grad is not None:
            p.grad.zero_()
    for n in range(iter_n):
        head = n * batch_interval
        tail = min((n + 1) * batch_interval, sample_count)
        batch_coord = data.coord_pool[head:tail:down_rate].contiguous()
        batch_label = data.sdf_label_pool[head:tail:down_rate].contiguous()
        perm, slots = plan_batch(octree, batch_coord)
        fused_train_step(octree, mlp, batch_coord, batch_label, None, opts, perm=perm, slots=slots)
        for i in range(len(octree.importance_weight)):
            g = _dense_grad(octree.hier_features[i])
            imp = octree.importance_weight[i]
            _lib.check(_lib.lib().shine_importance_accumulate(imp.data_ptr(), g.data_ptr(), imp.shape[0] - 1, stream),
                       "shine_importance_accumulate")
