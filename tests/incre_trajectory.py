"""Oracle side of the multi-frame incremental trajectory check (shine_incre.py:100-195): helper of
tests/test_gpu_parity.py::test_incremental_trajectory_matches_oracle and tests/test_oracle.py.

Per frame the reference runs
    dataset.process_frame -> octree.update(surface points, incremental_on=True)       shine_incre.py:105, lidar_dataset.py:215
    opt = setup_optimizer(...)                     a NEW Adam every frame              shine_incre.py:107-109
    iters x { query_feature, sdf, sdf_bce_loss(sum) + lambda_forget * cal_regularization, backward, opt.step }   :114-181
    cal_feature_importance(...)                                                         :190-195, utils/incre_learning.py:8-40
This module drives oracle/shine_oracle.py through exactly that sequence on batches and fresh feature rows handed in by
the caller (the GPU test hands in what the HIP path drew), in two modes:

  literal  so.train_step(regularize=True): the reference's op sequence.  From a level's second growth on,
           features_last_frame is an attached clone (model/feature_octree.py:160), so autograd adds +2 lambda imp (F - F_last)
           through one path and subtracts it through the other: the regulariser's gradient cancels, but in fp32 it leaves
           ~ulp(lambda imp diff) of rounding noise on every touched row, which Adam (eps = 1e-15: a normalised step) turns
           into O(lr) differences on elements whose true gradient is below that noise.
  clean    the same loss VALUE, the regulariser's gradient taken as what it is in exact arithmetic (live while the copy is
           detached — a level's first growth —, zero afterwards): what the HIP path computes (FeatureOctree._reg_grad_on).
"""
import torch

from oracle import shine_oracle as so


class OracleIncremental:
    def __init__(self, ocfg, lr=0.01, weight_decay=1e-7, literal=True, decoder_state=None):
        self.cfg = ocfg
        self.lr, self.wd = lr, weight_decay
        self.literal = literal
        self.octree = so.OracleOctree(ocfg)
        self.mlp = so.OracleDecoder(ocfg, generator=torch.Generator().manual_seed(1))
        if decoder_state is not None:
            self.mlp.load_state_dict(decoder_state)
        self.grad_on = [True] * ocfg.tree_level_feat  # the regulariser's gradient is live (detached copy) per level
        self.opt = None
        self.losses = []

    def begin_frame(self, surface_points, new_rows=None):
        """update(incremental_on=True) + a new optimiser.  new_rows: per level (top-down) the feature table the HIP path
        holds after ITS update (same shape): the oracle's own randn rows — the rows this update appended — are replaced by
        the product's, the rows that existed before stay the oracle's own (they carry its trajectory)."""
        oct_ = self.octree
        before = [t.shape[0] for t in oct_.hier_features]
        first = len(before) == 0
        oct_.update(surface_points, incremental_on=True)
        grew = []
        for fl, t in enumerate(oct_.hier_features):
            g = first or t.shape[0] != before[fl]
            grew.append(g)
            if not g:
                continue
            if new_rows is not None:
                src = new_rows[fl].detach().cpu().to(torch.float32)
                assert src.shape == t.shape, (fl, src.shape, t.shape)
                keep = 0 if first else before[fl] - 1  # (the old trash row is dropped by update, :156)
                oct_.hier_features[fl] = torch.cat((t.detach()[:keep], src[keep:]), 0).requires_grad_(True)
            if first:  # model/feature_octree.py:146: a detached copy
                oct_.features_last_frame[fl] = oct_.hier_features[fl].detach().clone()
                self.grad_on[fl] = True
            else:  # :160: clone of the Parameter itself, attached
                oct_.features_last_frame[fl] = oct_.hier_features[fl].clone()
                self.grad_on[fl] = False
        self.opt = so.adam_param_groups(oct_, self.mlp, lr=self.lr, weight_decay=self.wd)  # shine_incre.py:107-109
        return grew

    def freeze_decoder(self):
        """freeze_model(geo_mlp), utils/tools.py:188-191 (shine_incre.py:93-97): the decoder's tensors stop requiring grad; the
        per-frame optimiser still lists them (shine_incre.py:107-109) and Adam skips parameters without a gradient"""
        for p in self.mlp.params():
            p.requires_grad_(False)

    def iterate(self, coord, label, weight):
        oct_, mlp, cfg = self.octree, self.mlp, self.cfg
        if self.literal:
            out = so.train_step(oct_, mlp, coord, label, weight, cfg, regularize=True)
            loss = float(out["loss"])
        else:
            out = so.train_step(oct_, mlp, coord, label, weight, cfg, regularize=False)
            with torch.no_grad():
                reg = 0.0
                L = oct_.featured_level_num
                for i in range(L):
                    fl = L - i - 1
                    u = oct_.hierarchical_indices[i].flatten().unique()
                    diff = oct_.hier_features[fl][u] - oct_.features_last_frame[fl][u].detach()
                    imp = oct_.importance_weight[fl][u]
                    reg = reg + (imp * diff ** 2).sum()
                    if self.grad_on[fl]:
                        oct_.hier_features[fl].grad[u] += 2.0 * cfg.lambda_forget * imp * diff
            loss = float(out["loss"]) + cfg.lambda_forget * float(reg)
        self.opt.step()
        self.losses.append(loss)
        return loss

    def end_frame(self, coord_pool, label_pool, bs, down_rate=2):
        self.octree.zero_grad()  # opt.zero_grad(set_to_none=True), shine_incre.py:191
        self.mlp.zero_grad()
        so.importance_sweep(self.octree, self.mlp, coord_pool, label_pool, self.cfg, bs, down_rate)

    def state(self):
        return dict(features=[t.detach().clone() for t in self.octree.hier_features],
                    decoder=[p.detach().clone() for p in self.mlp.params()],
                    importance=[t.detach().clone() for t in self.octree.importance_weight],
                    features_last=[t.detach().clone() for t in self.octree.features_last_frame])


def deviation(a, b, tol=1e-4):
    """(max |a - b| / max |b|, number of elements further than tol * max |b| apart)"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-30)
    d = (a - b).abs()
    return float(d.max()) / scale, int((d > tol * scale).sum())
