"""Decoder — host-side mirror of model/decoder.py:9-101 (geo decoder; `sdf` is the hot-path method).

Same constructor signature, same sub-module names (so ``pretrained/geo_decoder_8dim.pth`` and the
reference's checkpoints load with ``load_state_dict``): ``layers.{0,1}``, ``lout``, ``nclass_out``.
`sdf` is a twice-differentiable HIP op (autograd_ops.FusedMLP: forward, backward and backward-of-backward are
one launch each — get_gradient(create_graph=True), utils/tools.py:175-185, differentiates it twice) for the
decoder shape every shipped config uses (8 -> 32 -> 32 -> 1 with bias) on CUDA float32 input; the benchmarked
tier never calls it — the fused HIP step reads the six parameter tensors directly (ops.train_step).
The out-of-scope heads (time-conditioned, semantic; flags off in every shipped yaml) and other decoder shapes
are plain torch composites, as in the reference.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ext, _lib, autograd_ops
from .autograd_ops import FusedInterpSdf, FusedMLP


class Decoder(nn.Module):
    def __init__(self, config, is_geo_encoder=True, is_time_conditioned=False):
        super().__init__()
        if is_geo_encoder:
            hidden, bias_on, level = config.geo_mlp_hidden_dim, config.geo_mlp_bias_on, config.geo_mlp_level
        else:
            hidden, bias_on, level = config.sem_mlp_hidden_dim, config.sem_mlp_bias_on, config.sem_mlp_level
        in_dim = config.feature_dim + (1 if is_time_conditioned else 0)
        self.layers = nn.ModuleList(
            [nn.Linear(in_dim if i == 0 else hidden, hidden, bias_on) for i in range(level)]
        )
        self.lout = nn.Linear(hidden, 1, bias_on)
        self.nclass_out = nn.Linear(hidden, config.sem_class_count + 1, bias_on)
        self.fusable = (
            level == 2 and hidden == _lib.HIDDEN_DIM and in_dim == _lib.FEATURE_DIM and bias_on
            and not is_time_conditioned
        )
        self.to(config.device)

    def forward(self, feature):
        return self.sdf(feature)

    # model/decoder.py:49-63
    def sdf(self, sum_features):
        if self.fusable and sum_features.is_cuda and sum_features.dtype == torch.float32 and sum_features.dim() == 2:
            mlp = self.fused_params()
            src = getattr(sum_features, "_shine_src", None)
            # (the six tensors query_feature's launch already evaluated this decoder on were checked there, a moment ago)
            spec = src.speculated(self, mlp) if src is not None else None
            if spec is None and not self._params_on(sum_features.device, mlp):
                return self._sdf_composite(sum_features)
            if src is not None and torch.is_grad_enabled() and sum_features.requires_grad and src.fusable(sum_features):
                # the untouched output of FeatureOctree.query_feature (shine_batch.py:123-124): interpolation + decoder as
                # ONE autograd node whose backward is one fused launch
                octree = src.octree
                octree.__dict__["_spec_decoder"] = weakref.ref(self)  # (the next query_feature evaluates this decoder too)
                ext = _ext.module()
                feats = octree.feature_list()
                if ext is not None and src.coord.is_cuda:  # the C++ node (csrc/shine_torch_ext.cpp)
                    octree._require_tables(with_ranks=True, probe=False)  # (its backward plans the batch: node ranks)
                    pred, link = ext.fused_sdf(octree._ext_state(ext), sum_features, src.coord, spec, feats, mlp,
                                               autograd_ops.DETERMINISTIC_BACKWARD)
                    pred._shine_link = (src, tuple(feats) + tuple(mlp), link)
                    return pred
                pred = FusedInterpSdf.apply(sum_features.detach(), src.coord, octree, src, spec, *feats, *mlp)
                pred._shine_link = (src, tuple(feats) + tuple(mlp), None)
                return pred
            return FusedMLP.apply(sum_features, *mlp)
        return self._sdf_composite(sum_features)

    def _sdf_composite(self, sum_features):
        h = sum_features  # other shapes / devices: the reference's composite
        for l in self.layers:
            h = F.relu(l(h))
        return self.lout(h).squeeze(1)

    # model/decoder.py:65-81
    def time_conditionded_sdf(self, sum_features, ts):
        h = torch.cat((sum_features, ts.view(-1, 1)), dim=1)
        for l in self.layers:
            h = F.relu(l(h))
        return self.lout(h).squeeze(1)

    # model/decoder.py:84-86
    def occupancy(self, sum_features):
        return torch.sigmoid(self.sdf(sum_features))

    # model/decoder.py:89-101
    def sem_label_prob(self, sum_features):
        h = sum_features
        for l in self.layers:
            h = F.relu(l(h))
        return F.log_softmax(self.nclass_out(h), dim=1)

    def sem_label(self, sum_features):
        return torch.argmax(self.sem_label_prob(sum_features), dim=1)

    # ---- fused-path plumbing
    def _params_on(self, device, mlp=None) -> bool:
        """the HIP decoder reads the six tensors in place: CUDA float32 contiguous on the input's device, or the composite runs"""
        return all(p.is_cuda and p.device == device and p.dtype == torch.float32 and p.is_contiguous()
                   for p in (mlp if mlp is not None else self.fused_params()))

    def fused_params(self):
        """W1,b1,W2,b2,w3,b3 in the order libshine_hip expects."""
        if not self.fusable:
            raise NotImplementedError("fused HIP decoder supports 8 -> 32 -> 32 -> 1 with bias (every shipped config)")
        # (plain dict lookups: nn.Module.__getattr__ / ModuleList.__getitem__ cost ~1 us per hop, and the small-batch loop asks
        # for these six tensors half a dozen times per iteration)
        m = self._modules
        hidden = m["layers"]._modules
        l0, l1, lo = hidden["0"]._parameters, hidden["1"]._parameters, m["lout"]._parameters
        return [l0["weight"], l0["bias"], l1["weight"], l1["bias"], lo["weight"], lo["bias"]]
