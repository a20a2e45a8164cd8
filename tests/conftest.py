import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_NAMES = ["maicity_bce_L3", "maicity_bce_L4", "kitti_eik_L3", "ncd_reg_L3", "linear_L2_nopoly"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), map_location="cpu", weights_only=False)


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return load_golden(request.param)


def oracle_from_golden(fx):
    """Rebuild the CPU oracle's octree + decoder from a fixture (tables, features, decoder weights)."""
    from oracle import shine_oracle as so

    cfg = so.make_config(**fx["cfg"])
    oct_ = so.OracleOctree(cfg)
    L = cfg.tree_level_feat
    for s, (keys, ids) in enumerate(fx["tables"]):
        lvl = oct_.free_level_num + s
        oct_.node_table[lvl] = dict(zip(keys.tolist(), ids.tolist()))
    oct_.hier_features = [f.clone().requires_grad_(True) for f in fx["features"]]
    if fx["regularize"]:
        oct_.importance_weight = [t.clone() for t in fx["importance"]]
        # fixtures store values only; the attached-clone quirk (feature_octree.py:160) is re-created here
        oct_.features_last_frame = [t.clone() for t in fx["features_last"]]
    mlp = so.OracleDecoder(cfg)
    mlp.load_state_dict(fx["decoder"])
    assert len(oct_.hier_features) == L
    return cfg, oct_, mlp


def product_from_golden(fx, device="cuda"):
    """Our FeatureOctree/Decoder loaded with a fixture's tables and weights."""
    from shine_mapping_amd import Decoder, FeatureOctree, synth

    cfg = synth.make_config("maicity", device=device, **fx["cfg"])
    octree = FeatureOctree(cfg)
    octree.load_tables(fx["tables"])
    for f in fx["features"]:
        octree.hier_features.append(torch.nn.Parameter(f.clone().to(device)))
    if fx["regularize"]:
        octree.importance_weight = [t.clone().to(device) for t in fx["importance"]]
        octree.features_last_frame = [t.clone().to(device) for t in fx["features_last"]]
    dec = Decoder(cfg)
    dec.load_state_dict(fx["decoder"], strict=False)
    return cfg, octree, dec


def oracle_from_product(octree, dec, cfg):
    """The CPU oracle loaded with a PRODUCT octree's tables / features and decoder weights (synthetic workloads built by
    shine_mapping_amd.synth): lets the at-scale GPU tests compare with the oracle on workloads no fixture holds."""
    from oracle import shine_oracle as so

    ocfg = so.make_config(tree_level_world=cfg.tree_level_world, tree_level_feat=cfg.tree_level_feat,
                          leaf_vox_size=cfg.leaf_vox_size, sigma_sigmoid_m=cfg.sigma_sigmoid_m,
                          ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, poly_int_on=cfg.poly_int_on,
                          loss_reduction=cfg.loss_reduction, lambda_forget=getattr(cfg, "lambda_forget", 0.0))
    oct_ = so.OracleOctree(ocfg)
    for lvl, tab in enumerate(octree.nodes_lookup_tables):
        oct_.node_table[lvl] = tab
    oct_.hier_features = [p.detach().cpu().clone().requires_grad_(True) for p in octree.hier_features]
    if len(octree.importance_weight):
        oct_.importance_weight = [t.detach().cpu().clone() for t in octree.importance_weight]
        oct_.features_last_frame = [t.detach().cpu().clone() for t in octree.features_last_frame]
    mlp = so.OracleDecoder(ocfg)
    mlp.load_state_dict({k: v.detach().cpu() for k, v in dec.state_dict().items()})
    return ocfg, oct_, mlp


def oracle_from_product_restricted(octree, dec, cfg, coord):
    """oracle_from_product for a map too large for a python dict of ALL its nodes (10^7): the oracle's node tables hold only the
    nodes the batch `coord` addresses — found on the oracle's side (its own quantise + Morton arithmetic, oracle/kaolin_shim.py)
    in the product's host copies of the (node key, corner ids) arrays (the tables the reference's update() would have built:
    test_device_octree_build_matches_reference_tables)."""
    import numpy as np

    from oracle import kaolin_shim as kal
    from oracle import shine_oracle as so

    ocfg = so.make_config(tree_level_world=cfg.tree_level_world, tree_level_feat=cfg.tree_level_feat,
                          leaf_vox_size=cfg.leaf_vox_size, sigma_sigmoid_m=cfg.sigma_sigmoid_m,
                          ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e, poly_int_on=cfg.poly_int_on,
                          loss_reduction=cfg.loss_reduction, lambda_forget=getattr(cfg, "lambda_forget", 0.0))
    oct_ = so.OracleOctree(ocfg)
    octree._sync_host()
    L = octree.featured_level_num
    c = coord.detach().cpu()
    for s in range(L):
        lvl = octree.free_level_num + s
        want = np.unique(kal.points_to_morton(kal.quantize_points(c, lvl)).cpu().numpy().astype(np.int64))
        keys, ids = octree._node_keys[s], octree._node_ids[s]
        order = np.argsort(keys, kind="stable")
        pos = np.searchsorted(keys[order], want)
        pos[pos >= keys.size] = keys.size - 1
        at = order[pos]
        hit = keys[at] == want
        oct_.node_table[lvl] = dict(zip(want[hit].tolist(), ids[at[hit]].tolist()))
    oct_.hier_features = [p.detach().cpu().clone().requires_grad_(True) for p in octree.hier_features]
    mlp = so.OracleDecoder(ocfg)
    mlp.load_state_dict({k: v.detach().cpu() for k, v in dec.state_dict().items()})
    return ocfg, oct_, mlp


def feat_grads_of_the_fused_terms(fx):
    """Feature grads of a fixture WITHOUT the regulariser term, from the CPU oracle (fp32, the reference's op sequence).

    The fixtures of the incremental configuration were recorded with the regulariser in the loss.  Its gradient cancels
    exactly in exact arithmetic (features_last_frame is an attached clone of the Parameter, model/feature_octree.py:160), but
    in fp32 the reference adds and subtracts lambda_forget = 1e4 times the importance and keeps ~1e-4..3e-4 (of max-abs) of
    rounding noise in its recorded grads.  The HIP path computes the fused terms only, so it is held to THIS clean value at
    the contract's 1e-4, and to the recorded grads at 1e-4 + the recorded grads' own distance from the clean value."""
    from oracle import shine_oracle as so

    ocfg, oct_, mlp = oracle_from_golden(fx)
    out = so.train_step(oct_, mlp, fx["coord"], fx["sdf_label"], fx["weight"], ocfg, regularize=False)
    return out["feat_grads"]
