"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the reference's golden vectors
and against the CPU oracle on seeded inputs.

Tolerances (north_star: "SDF-value and gradient match to the reference within 1e-4 fp32 on identical
sampled points"): hierarchical_indices exact; pred / feat / g / loss <= 1e-4 absolute-or-relative;
gradient tensors <= 1e-4 of the tensor's max-abs (fp32 atomics reorder the sums).
"""
import pytest
import torch
import numpy as np

from conftest import feat_grads_of_the_fused_terms, load_golden, oracle_from_golden, product_from_golden

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def abs_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def decoder_grad_errs(grads, refs):
    """rel_err per decoder tensor (W1, b1, W2, b2, w3, b3), each against its own max-abs — except the ONE-element b3, whose
    gradient sum_p delta_p can cancel to 1e-2 of its own summands' size (then fp32 summation order alone moves it by 1e-3 of
    itself, in the reference's sum as in ours): it is scaled by the output layer's gradients (w3, b3) taken together."""
    out = [rel_err(g, r) for g, r in zip(grads[:5], refs[:5])]
    scale = max(float(refs[4].abs().max()), float(refs[5].abs().max()), 1e-30)
    out.append(abs_err(grads[5], refs[5]) / scale)
    return out


def g_close(g, ref, tol):
    """g = sigma * d pred / d coord compared between two KERNELS (different fp32 summation orders of the same sums).
    g is discontinuous where a ReLU pre-activation crosses zero: a point whose pre-activation sits within fp32 rounding of
    the kink takes one branch in one summation order and the other branch in the other (pred, which is continuous, still
    agrees).  A handful of such points per 10^5 is the nature of the function (tests/test_gpu_scale_parity.py checks
    against the oracle that every such point IS at a kink); everything else must match to `tol` of max |g|."""
    a, b = g.detach().double().cpu(), ref.detach().double().cpu()
    err = (a - b).abs().max(dim=1).values / max(float(b.abs().max()), 1e-30)
    bad = int((err > tol).sum())
    return bad <= max(2, a.shape[0] // 20000)


def step_options(fx):
    from shine_mapping_amd import StepOptions

    c = fx["cfg"]
    return StepOptions(sigma=fx["sigma"], loss_reduction=c.get("loss_reduction", "mean"),
                       ekional_loss_on=c.get("ekional_loss_on", False), weight_e=c.get("weight_e", 0.1))


def test_forward_matches_reference(golden):
    from shine_mapping_amd import forward_sdf

    cfg, octree, dec = product_from_golden(golden)
    ref = golden["out"]
    coord = golden["coord"].cuda()
    out = forward_sdf(octree, dec, coord, want_feat=True, want_indices=True,
                      want_grad_x=ref["g"] is not None, sigma=golden["sigma"])
    torch.cuda.synchronize()
    for k in range(len(ref["indices"])):
        assert torch.equal(out["indices"][k].cpu(), ref["indices"][k]), "hierarchical_indices[%d] differ" % k
    assert abs_err(out["feat"], ref["feat"]) <= TOL
    assert abs_err(out["pred"], ref["pred"]) <= TOL
    if ref["g"] is not None:
        assert rel_err(out["grad_x"], ref["g"]) <= TOL


def test_get_indices_matches_reference(golden):
    cfg, octree, dec = product_from_golden(golden)
    idx = octree.get_indices(golden["coord"].cuda())
    for k, r in enumerate(golden["out"]["indices"]):
        assert torch.equal(idx[k].cpu(), r)


def test_fused_train_step_matches_reference(golden):
    from shine_mapping_amd import fused_train_step

    cfg, octree, dec = product_from_golden(golden)
    ref = golden["out"]
    loss, pred, g = fused_train_step(octree, dec, golden["coord"].cuda(), golden["sdf_label"].cuda(),
                                     golden["weight"].cuda(), step_options(golden), want_grad_x=True)
    torch.cuda.synchronize()
    assert abs_err(pred, ref["pred"]) <= TOL
    if ref["g"] is not None:
        assert rel_err(g, ref["g"]) <= TOL
    # loss: the regulariser term is a separate op (FeatureOctree.cal_regularization); compare the fused part
    expect = ref["parts"]["bce"].double()
    if "eikonal" in ref["parts"]:
        expect = expect + golden["cfg"]["weight_e"] * ref["parts"]["eikonal"].double()
    assert abs(float(loss) - float(expect)) <= TOL * max(1.0, abs(float(expect)))
    # with the regulariser the reference's RECORDED grads carry ~1e-4 of cancellation noise of their own: hold the HIP path to
    # the clean fused-term grads at TOL, and to the recorded ones at TOL + their distance from the clean value
    clean = feat_grads_of_the_fused_terms(golden) if golden["regularize"] else ref["feat_grads"]
    for k, (r, c) in enumerate(zip(ref["feat_grads"], clean)):
        assert rel_err(octree.hier_features[k].grad, c) <= TOL, "feature grad level %d" % k
        assert rel_err(octree.hier_features[k].grad, r) <= TOL + rel_err(r, c), "feature grad level %d (recorded)" % k
    for k, (p, r) in enumerate(zip(dec.fused_params(), ref["mlp_grads"])):
        assert rel_err(p.grad, r) <= TOL, "decoder grad %d" % k


def test_trash_row_gets_gradient_like_the_reference():
    fx = load_golden("maicity_bce_L3")
    from shine_mapping_amd import fused_train_step

    cfg, octree, dec = product_from_golden(fx)
    fused_train_step(octree, dec, fx["coord"].cuda(), fx["sdf_label"].cuda(), fx["weight"].cuda(), step_options(fx))
    for k, r in enumerate(fx["out"]["feat_grads"]):
        assert float(r[-1].abs().max()) > 0  # the reference does accumulate into the trash row
        assert rel_err(octree.hier_features[k].grad[-1], r[-1]) <= TOL


def test_frozen_decoder_gets_no_grad():
    fx = load_golden("maicity_bce_L3")
    from shine_mapping_amd import fused_train_step

    cfg, octree, dec = product_from_golden(fx)
    for p in dec.parameters():
        p.requires_grad = False  # utils/tools.py:188-191 freeze_model
    fused_train_step(octree, dec, fx["coord"].cuda(), fx["sdf_label"].cuda(), fx["weight"].cuda(), step_options(fx))
    assert all(p.grad is None for p in dec.parameters())
    for k, r in enumerate(fx["out"]["feat_grads"]):
        assert rel_err(octree.hier_features[k].grad, r) <= TOL


def test_empty_and_ragged_batches():
    fx = load_golden("maicity_bce_L3")
    from shine_mapping_amd import forward_sdf, fused_train_step

    cfg, octree, dec = product_from_golden(fx)
    e = torch.zeros((0, 3), device="cuda")
    out = forward_sdf(octree, dec, e)
    assert out["pred"].shape == (0,)
    loss, pred, _ = fused_train_step(octree, dec, e, torch.zeros(0, device="cuda"), torch.zeros(0, device="cuda"),
                                     step_options(fx))
    assert float(loss) == 0.0 and pred.shape == (0,)
    # ragged: N = 1 and N = 257 (not a multiple of the 64-lane wave or the 256-thread block)
    _, oct_, mlp = oracle_from_golden(fx)
    from oracle import shine_oracle as so
    ocfg = so.make_config(**fx["cfg"])
    for n in (1, 257):
        c, l, w = fx["coord"][:n], fx["sdf_label"][:n], fx["weight"][:n]
        ref = so.train_step(oct_, mlp, c, l, w, ocfg)
        for p in list(octree.hier_features) + list(dec.parameters()):
            p.grad = None
        loss, pred, _ = fused_train_step(octree, dec, c.cuda(), l.cuda(), w.cuda(), step_options(fx))
        assert abs_err(pred, ref["pred"]) <= TOL
        assert abs(float(loss) - float(ref["loss"])) <= TOL * max(1.0, abs(float(ref["loss"])))
        for k, r in enumerate(ref["feat_grads"]):
            assert rel_err(octree.hier_features[k].grad, r) <= TOL


def test_seeded_batch_against_oracle_with_grown_octree():
    """Our own update() (two frames -> hash growth + rehash) then a 4096-point step, vs the oracle fed the same tables."""
    from oracle import shine_oracle as so
    from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, fused_train_step, synth

    cfg = synth.make_config("kitti", device="cuda")
    torch.manual_seed(7)
    octree, dec = FeatureOctree(cfg), Decoder(cfg)
    frames = list(synth.make_frames(cfg, frames=3, beams=16, azimuths=180, seed=5, device="cuda"))
    for c, l, w in frames:
        octree.update(c[w > 0])
    with torch.no_grad():
        for p in octree.hier_features:
            p[:-1] *= 10.0
    pool = synth.SimpleNamespace(coord=torch.cat([f[0] for f in frames]), sdf_label=torch.cat([f[1] for f in frames]),
                                 weight=torch.cat([f[2] for f in frames]))
    g = torch.Generator(device="cuda").manual_seed(11)
    coord, label, weight = synth.draw_batch(pool, 4096, g)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=True, weight_e=0.1)
    loss, pred, gx = fused_train_step(octree, dec, coord, label, weight, opts, want_grad_x=True)

    ocfg = so.make_config(**{k: getattr(cfg, k) for k in ("tree_level_world", "tree_level_feat", "leaf_vox_size",
                                                         "sigma_sigmoid_m", "ekional_loss_on", "weight_e")})
    oct_ = so.OracleOctree(ocfg)
    for lvl, tab in enumerate(octree.nodes_lookup_tables):
        oct_.node_table[lvl] = tab
    oct_.hier_features = [p.detach().cpu().clone().requires_grad_(True) for p in octree.hier_features]
    mlp = so.OracleDecoder(ocfg)
    mlp.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
    ref = so.train_step(oct_, mlp, coord.cpu(), label.cpu(), weight.cpu(), ocfg)
    assert abs_err(pred, ref["pred"]) <= TOL
    assert rel_err(gx, ref["g"]) <= TOL
    assert abs(float(loss) - float(ref["loss"])) <= TOL * max(1.0, abs(float(ref["loss"])))
    for k, r in enumerate(ref["feat_grads"]):
        assert rel_err(octree.hier_features[k].grad, r) <= TOL
    for k, (p, r) in enumerate(zip(dec.fused_params(), ref["mlp_grads"])):
        assert rel_err(p.grad, r) <= TOL


def test_morton_sort_is_a_permutation_in_key_order_and_step_is_order_invariant():
    from oracle import kaolin_shim as kal
    from shine_mapping_amd import dp, fused_train_step

    fx = load_golden("maicity_bce_L4")
    cfg, octree, dec = product_from_golden(fx)
    coord = fx["coord"].cuda()
    perm = dp.morton_order(octree, coord)
    torch.cuda.synchronize()
    p = perm.cpu().long()
    assert torch.equal(torch.sort(p).values, torch.arange(coord.shape[0]))
    # the sort key is the Z-order code of the leaf voxel relative to the map's bounding box, clamped to it
    origin, bits = octree._sort_box()
    vox = kal.quantize_points(fx["coord"], cfg.tree_level_world).long() - torch.tensor(origin)
    vox = torch.minimum(vox.clamp(min=0), torch.tensor([(1 << b) - 1 for b in bits]))
    bmin = min(bits)
    keys = kal.points_to_morton((vox & ((1 << bmin) - 1)).short())
    sh = 3 * bmin
    for axis in (2, 1, 0):  # z_hi, then y_hi, then x_hi on top
        keys = keys | ((vox[:, axis] >> bmin) << sh)
        sh += bits[axis] - bmin
    assert sum(bits) <= 32
    assert bool((keys[p][1:] >= keys[p][:-1]).all())
    loss, pred, _ = fused_train_step(octree, dec, coord, fx["sdf_label"].cuda(), fx["weight"].cuda(), step_options(fx),
                                     perm=perm)
    ref = fx["out"]
    assert abs_err(pred, ref["pred"]) <= TOL
    for k, r in enumerate(ref["feat_grads"]):
        assert rel_err(octree.hier_features[k].grad, r) <= TOL


@pytest.mark.parametrize("levels,n", [(3, 1 << 14), (4, 1 << 18)])
def test_fused_step_against_the_reference_kernel_at_scale_and_grad_checksum(levels, n):
    """BASELINE-size batch (2^18 points, 4 levels): the fused MFMA / run-length step (the batch is planned automatically) vs
    the lane-per-point reference kernel of the check library on the same device inputs, plus a size-independent property: per level, the column sums of the feature-grad table
    (trash row included) equal sum_p d loss/d f_p up to rounding, because the 8 corner weights sum to 1 — so
    every level must report the SAME column sums ("checksum of checksums")."""
    from shine_mapping_amd import StepOptions, dp, fused_train_step, synth

    wl = synth.build_workload("maicity", frames=6, device="cuda", seed=3, tree_level_feat=levels, azimuths=300)
    cfg, octree, dec = wl.cfg, wl.octree, wl.decoder
    with torch.no_grad():
        for p in octree.hier_features:
            p[:-1] *= 6.0
    g = torch.Generator(device="cuda").manual_seed(5)
    coord, label, weight = synth.draw_batch(wl.pool, n, g)
    params = list(octree.hier_features) + dec.fused_params()
    outs = {}
    for variant, perm in ((1, None), (0, None), (0, dp.morton_order(octree, coord))):
        for p in params:
            p.grad = None
        opts = StepOptions(sigma=cfg.sigma_sigmoid, kernel_variant=variant)
        loss, pred, _ = fused_train_step(octree, dec, coord, label, weight, opts, perm=perm)
        torch.cuda.synchronize()
        outs[(variant, perm is not None)] = (float(loss), pred.clone(), [p.grad.clone() for p in params])
    ref = outs[(1, False)]
    for key in ((0, False), (0, True)):
        got = outs[key]
        assert abs(got[0] - ref[0]) <= 1e-5 * max(1.0, abs(ref[0])), key
        assert abs_err(got[1], ref[1]) <= 2e-5, key
        for a, b in zip(got[2], ref[2]):
            assert rel_err(a, b) <= TOL, key
    sums = [gr.double().sum(0) for gr in outs[(0, True)][2][:levels]]
    scale = max(float(s.abs().max()) for s in sums)
    for s in sums[1:]:
        assert float((s - sums[0]).abs().max()) <= 1e-3 * scale


def test_tier_a_drop_in_loop_matches_reference(golden):
    """The reference's inner loop, verbatim (shine_batch.py:119-209 / shine_incre.py:129-180), on OUR classes:
    query_feature -> sdf -> get_gradient(create_graph=True) -> sdf_bce_loss (+ regulariser, + eikonal) ->
    loss.backward().  Exercises the autograd ops incl. the double backward."""
    from shine_mapping_amd import get_gradient, sdf_bce_loss

    cfg, octree, dec = product_from_golden(golden)
    ref = golden["out"]
    c = golden["cfg"]
    eik = c.get("ekional_loss_on", False)
    sigma = golden["sigma"]
    if golden["regularize"]:
        # feature_octree.py:160: the stored copy is an attached clone of the Parameter
        vals = octree.features_last_frame
        octree.features_last_frame = [p.clone() for p in octree.hier_features]
        with torch.no_grad():
            for t, v in zip(octree.features_last_frame, vals):
                t.copy_(v)
    coord = golden["coord"].cuda()
    sdf_label, weight = golden["sdf_label"].cuda(), golden["weight"].cuda()
    if eik:
        coord.requires_grad_(True)
    feature = octree.query_feature(coord)
    pred = dec.sdf(feature)
    surface_mask = weight > 0
    if eik:
        g = get_gradient(coord, pred) * sigma
    cur_loss = 0.0
    w_abs = torch.abs(weight)
    cur_loss = cur_loss + sdf_bce_loss(pred, sdf_label, sigma, w_abs, False, c.get("loss_reduction", "mean"))
    if golden["regularize"]:
        cur_loss = cur_loss + c["lambda_forget"] * octree.cal_regularization()
    if eik:
        cur_loss = cur_loss + c["weight_e"] * ((1.0 - g[surface_mask].norm(2, dim=-1)) ** 2).mean()
    cur_loss.backward()
    torch.cuda.synchronize()
    for k in range(len(ref["indices"])):
        assert torch.equal(octree.hierarchical_indices[k].cpu(), ref["indices"][k])
    assert abs_err(pred, ref["pred"]) <= TOL
    assert abs(float(cur_loss.detach()) - float(ref["loss"])) <= TOL * max(1.0, abs(float(ref["loss"])))
    if eik:
        assert rel_err(g, ref["g"]) <= TOL
    # This loop runs the regulariser too (torch ops on the attached clone, like the reference): its gradient cancels only up to
    # fp32 rounding of terms lambda_forget = 1e4 times larger than what remains — on BOTH sides.  The budget is the contract's
    # TOL plus twice the recorded grads' own distance from the clean fused-term grads (the reference's noise and ours).
    clean = feat_grads_of_the_fused_terms(golden) if golden["regularize"] else ref["feat_grads"]
    for k, (r, cg) in enumerate(zip(ref["feat_grads"], clean)):
        assert rel_err(octree.hier_features[k].grad, r) <= TOL + 2 * rel_err(r, cg), "feature grad level %d" % k
    for k, (p, r) in enumerate(zip(dec.fused_params(), ref["mlp_grads"])):
        assert rel_err(p.grad, r) <= TOL, "decoder grad %d" % k


def test_octree_pickle_roundtrip_keeps_tables():
    """utils/tools.py:200-213 pickles the whole FeatureOctree module; the device tables are rebuilt on load."""
    import io

    fx = load_golden("maicity_bce_L3")
    cfg, octree, dec = product_from_golden(fx)
    buf = io.BytesIO()
    torch.save({"feature_octree": octree}, buf)
    buf.seek(0)
    oct2 = torch.load(buf, weights_only=False)["feature_octree"]
    idx = oct2.get_indices(fx["coord"].cuda())
    for k, r in enumerate(fx["out"]["indices"]):
        assert torch.equal(idx[k].cpu(), r)
    assert oct2.nodes_lookup_tables[cfg.tree_level_world] == octree.nodes_lookup_tables[cfg.tree_level_world]


@pytest.mark.parametrize("name", ["maicity_bce_L4", "kitti_eik_L3", "linear_L2_nopoly"])
def test_plan_batch_counting_sort_and_planned_step(name):
    """shine_plan_batch: perm is a permutation, points of one (deepest) node are adjacent, the slots agree with
    get_indices' hits, and the fused step fed (perm, slots) reproduces the reference — for the counting sort (forced on these small
    batches with kernel_variant 0x800: batches of <= 16384 points are not reordered by default) and for the default plan."""
    from shine_mapping_amd import dp, fused_train_step

    fx = load_golden(name)
    coord = fx["coord"].cuda()
    for variant in (0x800, 0):
        cfg, octree, dec = product_from_golden(fx)
        perm, slots = dp.plan_batch(octree, coord, _debug_variant=variant, sort=False)  # (variant 0x800 asks for the sort)
        torch.cuda.synchronize()
        n, L = coord.shape[0], cfg.tree_level_feat
        p = perm.cpu().long()
        assert torch.equal(torch.sort(p).values, torch.arange(n))
        sl = slots.cpu()
        assert sl.shape == (n, L)
        ref_idx = fx["out"]["indices"]  # bottom-up; slots are top-down
        for s in range(L):
            hit_ref = (ref_idx[L - 1 - s][:, 0] >= 0)[p]
            assert torch.equal(sl[:, s] >= 0, hit_ref), "level slot %d hit pattern" % s
        if variant or n > 16384:  # points that share their deepest node are contiguous in the visiting order
            deepest = torch.full((n,), -1, dtype=torch.int64)
            for s in range(L):
                deepest = torch.where(sl[:, s] >= 0, sl[:, s].long() + (s << 40), deepest)
            change = (deepest[1:] != deepest[:-1]).sum().item() + 1
            assert change == len(torch.unique(deepest)), "a node's points are split into several runs"
        else:
            assert torch.equal(p, torch.arange(n))
        loss, pred, g = fused_train_step(octree, dec, coord, fx["sdf_label"].cuda(), fx["weight"].cuda(), step_options(fx),
                                         want_grad_x=True, perm=perm, slots=slots)
        ref = fx["out"]
        assert abs_err(pred, ref["pred"]) <= TOL
        if ref["g"] is not None:
            assert rel_err(g, ref["g"]) <= TOL
        for k, r in enumerate(ref["feat_grads"]):
            assert rel_err(octree.hier_features[k].grad, r) <= TOL
        for k, (pp, r) in enumerate(zip(dec.fused_params(), ref["mlp_grads"])):
            assert rel_err(pp.grad, r) <= TOL


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 1000, 4096, 16383, 16384, 16385])
def test_small_batches_are_planned_in_one_launch_and_not_reordered(n):
    """Batches of <= 16384 points (the reference's batch size is 4096) are not sorted (k_plan_unsorted: the node order buys a one-tile-per-wave
    step nothing and its histogram runs over every node of the tree): perm is the identity, every point's hash slots are the ones the
    counting sort finds, the ride-along clear of the gradient bucket still happens, and the plan is deterministic.  16385 points: the
    counting sort."""
    from shine_mapping_amd import dp

    fx = load_golden("kitti_eik_L3")
    cfg, octree, dec = product_from_golden(fx)
    g = torch.Generator().manual_seed(n)
    base = fx["coord"]
    coord = base[torch.randint(0, base.shape[0], (n,), generator=g)].clone()
    coord[::7] += 0.9  # some points that miss everywhere
    coord = coord.cuda()
    perm, slots = dp.plan_batch(octree, coord, sort=False)  # the per-iteration form of the step's own callers
    sperm, sslots = dp.plan_batch(octree, coord)            # the public default: node order at any size
    torch.cuda.synchronize()
    p, sl, sp, ssl = perm.cpu().long(), slots.cpu(), sperm.cpu().long(), sslots.cpu()
    assert torch.equal(torch.sort(p).values, torch.arange(n)) and torch.equal(torch.sort(sp).values, torch.arange(n))
    assert torch.equal(p, torch.arange(n)) == (n <= 16384)
    by_point, s_by_point = torch.empty_like(sl), torch.empty_like(ssl)
    by_point[p] = sl
    s_by_point[sp] = ssl
    assert torch.equal(by_point, s_by_point)
    for floats in (4, 1 << 20, (1 << 26) + 4):  # the gradient bucket rides on the launch
        flat = torch.ones(floats, device="cuda")
        perm3, slots3 = dp.plan_batch(octree, coord, zero=flat, sort=False)
        assert float(flat.abs().sum()) == 0.0
        if n <= 16384:
            assert torch.equal(perm3.cpu().long(), p) and torch.equal(slots3.cpu(), sl)


@pytest.mark.parametrize("attached", [False, True])
def test_regulariser_cpp_node_equals_python_node_and_reference_composite(attached, monkeypatch):
    """FeatureOctree.cal_regularization (model/feature_octree.py:246-255) through the C++ extension, the Python node and the
    reference's composite (unique + gathers) on the same query: same value; with a DETACHED features_last_frame (first frame) the
    same gradient 2 * importance * (F - F_last) on the query's rows; with the reference's attached clone (:160, later frames) the
    nodes leave no gradient at all (the composite's cancels up to rounding) and the C++ path does not even create a node."""
    fx = load_golden("ncd_reg_L3")
    coord = fx["coord"].cuda()
    res = {}
    for mode in ("ext", "python", "composite"):
        monkeypatch.setenv("SHINE_TIER_A_EXT", "1" if mode == "ext" else "0")
        cfg, octree, dec = product_from_golden(fx)
        vals = [v.detach() for v in octree.features_last_frame]
        octree.features_last_frame = vals
        if attached:
            octree.features_last_frame = [p.clone() for p in octree.hier_features]
            with torch.no_grad():
                for t, v in zip(octree.features_last_frame, vals):
                    t.copy_(v)
        octree.query_feature(coord)
        if mode == "composite":
            octree.hierarchical_indices  # (materialised: the composite path)
            reg = octree._cal_regularization_composite()
        else:
            reg = octree.cal_regularization()
            reg2 = octree.cal_regularization()  # (the flags are clean again: the second call sees the same rows)
            assert float(reg2) == float(reg)
        name = reg.grad_fn.name() if reg.grad_fn is not None else None
        if reg.requires_grad:
            (reg * 3.0).backward()
        torch.cuda.synchronize()
        res[mode] = (float(reg), [None if p.grad is None else p.grad.clone() for p in octree.hier_features], name)
    v_ref = res["composite"][0]
    assert v_ref > 0
    for mode in ("ext", "python"):
        assert abs(res[mode][0] - v_ref) <= 1e-5 * v_ref, (mode, res[mode][0], v_ref)
    assert res["ext"][2] == (None if attached else "OctreeRegularizer[ext]"), res["ext"][2]
    assert "OctreeRegularizer" in res["python"][2] and "[ext]" not in res["python"][2]
    for k in range(len(res["composite"][1])):
        g_ref = res["composite"][1][k]
        for mode in ("ext", "python"):
            g = res[mode][1][k]
            if attached:
                assert g is None or float(g.abs().max()) == 0.0
            else:
                assert rel_err(g, g_ref) <= 1e-5, (mode, k)
    if attached:  # the composite's own gradient is rounding noise around zero
        scale = 3.0 * 2.0 * max(float((i * (p - l).abs()).max()) for i, p, l in zip(
            octree.importance_weight, octree.hier_features, octree.features_last_frame))
        assert all(float(g.abs().max()) <= 1e-5 * max(scale, 1e-30) for g in res["composite"][1])


def test_fused_regulariser_and_importance_sweep_match_reference():
    """config ncd_incre_reg: fused step (sum reduction) + shine_regularize on the touched rows reproduces the
    reference loss (BCE + lambda * reg) and grads; cal_feature_importance (fused) reproduces the oracle's sweep."""
    from oracle import shine_oracle as so
    from shine_mapping_amd import dp, fused_train_step
    from shine_mapping_amd.incre_learning import cal_feature_importance
    from shine_mapping_amd.ops import fused_regularization, touched_flags

    fx = load_golden("ncd_reg_L3")
    cfg, octree, dec = product_from_golden(fx)
    octree._reg_grad_on = [False] * cfg.tree_level_feat  # fixture = second frame: attached clones (:160), value only
    coord, label, weight = fx["coord"].cuda(), fx["sdf_label"].cuda(), fx["weight"].cuda()
    touched = touched_flags(octree)
    perm, slots = dp.plan_batch(octree, coord)
    loss, pred, _ = fused_train_step(octree, dec, coord, label, weight, step_options(fx), perm=perm, slots=slots,
                                     touched=touched)
    reg = fused_regularization(octree, fx["cfg"]["lambda_forget"], touched)
    torch.cuda.synchronize()
    ref = fx["out"]
    assert abs(float(reg) - float(ref["parts"]["reg"])) <= 1e-4 * abs(float(ref["parts"]["reg"]))
    total = float(loss) + fx["cfg"]["lambda_forget"] * float(reg)
    assert abs(total - float(ref["loss"])) <= 1e-4 * abs(float(ref["loss"]))
    for k, r in enumerate(ref["feat_grads"]):
        assert rel_err(octree.hier_features[k].grad, r) <= 3e-4
    assert all(int(t.sum()) == 0 for t in touched)  # flags are cleared for the next iteration
    # with a detached copy (first-frame branch, :146) the regulariser DOES contribute: d/dF = 2*lambda*imp*(F-F_last)
    octree._reg_grad_on = [True] * cfg.tree_level_feat
    for p in octree.hier_features:
        p.grad = None
    fused_train_step(octree, dec, coord, label, weight, step_options(fx), perm=perm, slots=slots, touched=touched)
    base = [p.grad.clone() for p in octree.hier_features]
    idx = octree.get_indices(coord)
    fused_regularization(octree, fx["cfg"]["lambda_forget"], touched)
    L = cfg.tree_level_feat
    for s in range(L):
        u = idx[L - 1 - s].flatten().unique()
        u = u[u >= 0]
        expect = torch.zeros_like(base[s])
        d = octree.hier_features[s].detach()[u] - octree.features_last_frame[s][u]
        expect[u] = 2.0 * fx["cfg"]["lambda_forget"] * octree.importance_weight[s][u] * d
        assert rel_err(octree.hier_features[s].grad - base[s], expect) <= 1e-5

    # importance sweep vs the oracle restatement of utils/incre_learning.py:8-40
    ocfg, oct_, mlp = oracle_from_golden(fx)
    pool_c, pool_l = fx["coord"], fx["sdf_label"]
    for t in oct_.importance_weight:
        t.zero_()
    so.importance_sweep(oct_, mlp, pool_c, pool_l, ocfg, 256, 2)
    for t in octree.importance_weight:
        t.zero_()
    data = type("Pool", (), {"coord_pool": pool_c.cuda(), "sdf_label_pool": pool_l.cuda()})()
    cal_feature_importance(data, octree, dec, fx["sigma"], 256, 2, "sum")
    torch.cuda.synchronize()
    for a, b in zip(octree.importance_weight, oct_.importance_weight):
        assert rel_err(a, b) <= TOL
        assert float(a[-1].abs().max()) == 0.0


def test_device_update_appends_rows_like_the_torch_loop():
    """FeatureOctree.update on the device, second and later frames: the feature-side appends of all levels are ONE launch
    (shine_append_rows) and what the device added is fetched in ONE copy (shine_tables_grow_fetch_all) — against the per-level
    torch form of the same update (model/feature_octree.py:147-160) on the same generator state and the per-level fetch."""
    import ctypes as C

    from shine_mapping_amd import FeatureOctree, _lib, synth

    fx = load_golden("ncd_reg_L3")
    cfg = synth.make_config("maicity", device="cuda", **fx["cfg"])
    frames = [sp.cuda() for sp in fx["surface_points"]]
    assert len(frames) >= 2
    frames = frames + [frames[-1] * 0.97 + 0.011]  # one more frame that certainly adds nodes

    def run(fused):
        torch.manual_seed(11)
        octree = FeatureOctree(cfg)
        if not fused:  # the torch form, level by level
            octree._append_rows_fused = lambda levels, added, inc, dev, stream: [
                octree._append_rows(s, False, a, inc, dev) for s, a in zip(levels, added)]
        logs = []
        for f in frames:
            octree.update(f, incremental_on=True)
            # the per-level fetch of what this frame added, for comparison with the flat copy
            t = octree._tables
            _, nf, na = octree._dev_frames[-1]
            per_level = []
            for s in range(octree.featured_level_num):
                keys = torch.empty(nf[s], dtype=torch.int64, device="cuda")
                ids = torch.empty((nf[s], 8), dtype=torch.int32, device="cuda")
                newc = torch.empty(na[s], dtype=torch.int64, device="cuda")
                if nf[s]:
                    _lib.check(_lib.lib().shine_tables_grow_fetch(t.handle, s, keys.data_ptr(), ids.data_ptr(), newc.data_ptr(),
                                                                   _lib.current_stream_handle()), "fetch")
                per_level.append((keys, ids, newc))
            logs.append(per_level)
            with torch.no_grad():  # something for importance / last-frame copies to carry over
                for w in octree.importance_weight:
                    w[:-1] += 0.25
        torch.cuda.synchronize()
        return octree, logs

    a, logs = run(True)
    b, _ = run(False)
    for x, y in zip(a.hier_features, b.hier_features):
        assert x.shape == y.shape and torch.equal(x.detach(), y.detach()) and float(x[-1].abs().max()) == 0.0
    for x, y in zip(a.importance_weight, b.importance_weight):
        assert torch.equal(x, y)
    for x, y in zip(a.features_last_frame, b.features_last_frame):
        assert torch.equal(x.detach(), y.detach())
    assert all(t.grad_fn is not None for t in a.features_last_frame)  # the reference's attached clone (:160)
    assert a._reg_grad_on == b._reg_grad_on
    # the flat copies, split per level, are the per-level fetches
    frames_flat = list(a._dev_frames)
    a._drain_dev_frames()
    assert len(frames_flat) == len(logs)
    for s in range(a.featured_level_num):
        got = a._dev_log[s]
        want = [lv[s] for lv in logs if lv[s][0].numel()]
        assert len(got) == len(want)
        for (k1, i1, c1), (k2, i2, c2) in zip(got, want):
            assert torch.equal(k1, k2) and torch.equal(i1, i2) and torch.equal(c1, c2)


@pytest.mark.parametrize("n,bs,down_rate", [(1, 4, 1), (1000, 64, 1), (1001, 100, 3), (70000, 4096, 2), (4096, 4096, 1),
                                            (300001, 512, 5), (100000, 16384, 1), (100001, 5000, 3),
                                            (200000, 20000, 2)])  # (the last one: chunks beyond a workgroup's LDS, radix path)
def test_importance_chunks_match_the_torch_partition(n, bs, down_rate):
    """shine_importance_chunks (place + per-chunk LDS sort; one radix pass over the chunk ids for bs > 16384) == the torch form of the same partition: members of chunk c =
    pool[c * bs * down_rate : (c + 1) * bs * down_rate : down_rate] (utils/incre_learning.py:27-31), as ascending sorted positions."""
    import ctypes as C
    import math

    from shine_mapping_amd import _lib
    from shine_mapping_amd.incre_learning import chunk_partition

    g = torch.Generator().manual_seed(n)
    perm = torch.randperm(n, generator=g).to(torch.int32).cuda()
    interval = bs * down_rate
    iter_n = math.ceil(n / interval)
    want_idx, want_begin = chunk_partition(perm, n, interval, down_rate)
    lib = _lib.lib()
    idx = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    begin = (C.c_int64 * (iter_n + 1))()
    need = C.c_size_t()
    _lib.check(lib.shine_importance_chunks(None, n, bs, down_rate, None, None, iter_n, None, C.byref(need), None), "sizes")
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    _lib.check(lib.shine_importance_chunks(perm.data_ptr(), n, bs, down_rate, idx.data_ptr(), begin, iter_n, ws.data_ptr(),
                                           C.byref(need), _lib.current_stream_handle()), "shine_importance_chunks")
    torch.cuda.synchronize()
    assert list(begin) == want_begin
    kept = want_begin[-1]
    assert torch.equal(idx[:kept], want_idx[:kept])
    assert bool((idx[kept:] == -7).all())  # nothing is written behind the kept samples
    assert lib.shine_importance_chunks(perm.data_ptr(), n, bs, down_rate, idx.data_ptr(), begin, iter_n + 1, ws.data_ptr(),
                                       C.byref(need), None) == -1


@pytest.mark.parametrize("reduction,bs,down_rate,take,budget", [
    ("mean", 100, 3, 1001, None), ("sum", 4096, 1, 700, None), ("mean", 64, 2, 1024, None),
    ("mean", 4, 2, 1024, None),   # 128 chunks: two launches of 64
    ("sum", 64, 1, 1000, 3.5),    # a scratch budget of 3.5 chunks' tables: 16 chunks in groups of 3 (the last group has one)
    ("mean", 50, 1, 1000, 0.5)])  # less than one chunk's tables: one chunk per launch
def test_importance_sweep_chunking_matches_oracle(reduction, bs, down_rate, take, budget, monkeypatch):
    """cal_feature_importance's chunk loop runs behind the ABI (shine_importance_sweep): a chunk's gradient is summed
    before the abs, so chunk MEMBERSHIP (head:tail:down_rate of the pool in its original order, a short last chunk, one
    chunk larger than the pool) must be the reference's — utils/incre_learning.py:27-40 — as must the per-chunk 'mean'.
    Up to 64 chunks are ONE launch (each with its own gradient tables in the sweep's scratch); more chunks, or a scratch
    budget that holds fewer tables, take several."""
    import copy

    from oracle import shine_oracle as so
    from shine_mapping_amd import incre_learning
    from shine_mapping_amd.incre_learning import cal_feature_importance

    fx = load_golden("ncd_reg_L3")
    cfg, octree, dec = product_from_golden(fx)
    ocfg, oct_, mlp = oracle_from_golden(fx)
    ocfg = copy.copy(ocfg)
    ocfg.loss_reduction = reduction
    pool_c, pool_l = fx["coord"][:take].contiguous(), fx["sdf_label"][:take].contiguous()
    for t in oct_.importance_weight:
        t.zero_()
    so.importance_sweep(oct_, mlp, pool_c, pool_l, ocfg, bs, down_rate)
    for t in octree.importance_weight:
        t.zero_()
    data = type("Pool", (), {"coord_pool": pool_c.cuda(), "sdf_label_pool": pool_l.cuda()})()
    if budget is not None:
        per_chunk = sum(p.shape[0] * 32 + ((p.shape[0] + 15) & ~15) for p in octree.hier_features)
        monkeypatch.setattr(incre_learning, "SCRATCH_BUDGET_BYTES", int(budget * per_chunk))
    for p in octree.hier_features:  # (a leftover gradient must not leak into the first chunk: shine_incre.py:192 clears it)
        p.grad = torch.ones_like(p)
    cal_feature_importance(data, octree, dec, fx["sigma"], bs, down_rate, reduction)
    torch.cuda.synchronize()
    for a, b in zip(octree.importance_weight, oct_.importance_weight):
        assert rel_err(a, b) <= TOL
        assert float(a[-1].abs().max()) == 0.0
    for f in octree.hier_features:  # the sweep leaves the gradients cleared (incre_learning.py:38)
        assert float(f.grad.abs().max()) == 0.0
    # ... and its scratch (the chunks' private tables and row flags) zero, for the next call
    assert all(int(b.count_nonzero()) == 0 for b in incre_learning._SCRATCH.values())
    # the same sweep re-using the plan of the frame's SortedPool (what the incremental loop has in hand) instead of planning
    # the pool a second time: chunk membership comes from the pool's own permutation
    from shine_mapping_amd.sampler import SortedPool

    first = [t.clone() for t in octree.importance_weight]
    for canonical in (False, True):
        octree._require_tables(with_ranks=True)
        sp = SortedPool(octree, data.coord_pool, data.sdf_label_pool, torch.ones_like(data.sdf_label_pool), seed=1,
                        canonical=canonical)
        for t in octree.importance_weight:
            t.zero_()
        cal_feature_importance(data, octree, dec, fx["sigma"], bs, down_rate, reduction, pool=sp)
        torch.cuda.synchronize()
        for a, b in zip(octree.importance_weight, first):
            assert rel_err(a, b) <= 1e-5


def test_fused_adam_matches_torch_adam():
    """shine_adam_step vs torch.optim.Adam with the reference's groups (utils/tools.py:57-83): decoder with L2
    weight decay, one group per feature level, betas (0.9, 0.99), eps 1e-15; five steps, grads cleared in-pass."""
    from shine_mapping_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(3)
    shapes = [(32, 8), (32,), (32, 32), (32,), (1, 32), (1,), (1001, 8), (4003, 8), (16385, 8)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    groups = lambda t: [{"params": t[:6], "lr": 0.01, "weight_decay": 1e-7}, {"params": [t[8]], "lr": 0.01},
                        {"params": [t[7]], "lr": 0.005}, {"params": [t[6]], "lr": 0.0025}]
    ref = torch.optim.Adam(groups(qs), betas=(0.9, 0.99), eps=1e-15)
    opt = FusedAdam(groups(ps), betas=(0.9, 0.99), eps=1e-15)
    for it in range(5):
        for p, q in zip(ps, qs):
            gr = torch.randn(p.shape, generator=g).cuda() * (10.0 ** (it - 2))
            if p.dim() == 2 and p.shape[0] > 100:
                gr[::3] = 0.0  # untouched rows still move (dense momentum), like the reference
            p.grad = gr.clone()
            q.grad = gr.clone()
        ref.step()
        opt.step(zero_grad=True)
        for p, q in zip(ps, qs):
            assert rel_err(p, q) <= 2e-6
            assert float(p.grad.abs().max()) == 0.0


def test_active_row_adam_is_exact_against_torch_adam():
    """VERDICT r03 item 2b — exact active-row Adam (shine_adam_step row_flags): a sparse-touch schedule on the reference's
    optimiser groups (utils/tools.py:57-83) against the DENSE torch.optim.Adam.  Rows that never received a gradient are not
    read by the fused step and must equal torch's result bit for bit (torch leaves them unchanged: m = v = g = 0 gives
    p -= lr * 0 / (0 + eps)); rows touched at least once keep receiving dense updates (momentum on later untouched steps) and
    match to 2e-6; the flags end as 0 / 2 and flag 0 <=> exp_avg_sq == 0."""
    from shine_mapping_amd.optim import FusedAdam

    g = torch.Generator().manual_seed(5)
    shapes = [(32, 8), (32,), (32, 32), (32,), (1, 32), (1,), (1001, 8), (4003, 8), (70001, 8)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    start = [p.detach().clone() for p in ps]
    groups = lambda t: [{"params": t[:6], "lr": 0.01, "weight_decay": 1e-7}, {"params": [t[8]], "lr": 0.01},
                        {"params": [t[7]], "lr": 0.005}, {"params": [t[6]], "lr": 0.0025}]
    ref = torch.optim.Adam(groups(qs), betas=(0.9, 0.99), eps=1e-15)
    opt = FusedAdam(groups(ps), betas=(0.9, 0.99), eps=1e-15)
    flags = {p: torch.zeros(p.shape[0], dtype=torch.uint8, device="cuda") for p in ps[6:]}
    ever = {p: torch.zeros(p.shape[0], dtype=torch.bool, device="cuda") for p in ps[6:]}
    for it in range(6):
        for p, q in zip(ps, qs):
            gr = torch.randn(p.shape, generator=g).cuda() * (10.0 ** (it - 2))
            if p in flags:  # a sparse batch: ~3 % of the rows get a gradient (a different 3 % every iteration)
                hit = (torch.rand(p.shape[0], generator=g) < 0.03).cuda()
                gr = gr * hit[:, None]
                flags[p][hit] = 1  # what the fused step's scatter does (shine_train_step `touched`)
                ever[p] |= hit
            p.grad = gr.clone()
            q.grad = gr.clone()
        ref.step()
        opt.step(zero_grad=True, row_flags=flags)
    torch.cuda.synchronize()
    for p, q, s0 in zip(ps, qs, start):
        assert float(p.grad.abs().max()) == 0.0
        if p in flags:
            never = ~ever[p]
            assert int(never.sum()) > 0.5 * p.shape[0]
            assert torch.equal(p.detach()[never], q.detach()[never]) and torch.equal(p.detach()[never], s0[never])
            assert rel_err(p.detach()[ever[p]], q.detach()[ever[p]]) <= 2e-6
            assert torch.equal(flags[p] != 0, ever[p]) and int((flags[p] == 1).sum()) == 0
            m, v = opt.state[p]
            assert float(m[never].abs().max()) == 0.0 and float(v[never].abs().max()) == 0.0
            assert bool((v[ever[p]].abs().sum(dim=1) > 0).all())
            assert rel_err(m, ref.state[q]["exp_avg"]) <= 2e-6 and rel_err(v, ref.state[q]["exp_avg_sq"]) <= 2e-6
        else:
            assert rel_err(p, q) <= 2e-6


@pytest.mark.parametrize("name", ["maicity_bce_L3", "kitti_eik_L3"])
def test_training_trajectory_matches_oracle(name):
    """Five full iterations of the inner loop (shine_batch.py:105-210): plan -> fused step -> fused Adam on the GPU vs
    query -> sdf -> loss -> backward -> torch.optim.Adam in the CPU oracle, same batches, same optimiser groups
    (utils/tools.py:57-83).  The loss sequence and the final parameters must agree."""
    from oracle import shine_oracle as so
    from shine_mapping_amd import dp, fused_train_step
    from shine_mapping_amd.optim import setup_optimizer

    fx = load_golden(name)
    cfg, octree, dec = product_from_golden(fx)
    ocfg, oct_, mlp = oracle_from_golden(fx)
    cfg.lr, cfg.weight_decay, cfg.lr_level_reduce_ratio, cfg.adam_eps, cfg.opt_adam = 0.01, 1e-7, 1.0, 1e-15, True
    opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
    ref_opt = so.adam_param_groups(oct_, mlp, lr=0.01, weight_decay=1e-7)
    g = torch.Generator().manual_seed(9)
    n = fx["coord"].shape[0]
    losses, ref_losses = [], []
    for it in range(5):
        idx = torch.randint(0, n, (1024,), generator=g)
        c, l, w = fx["coord"][idx], fx["sdf_label"][idx], fx["weight"][idx]
        out = so.train_step(oct_, mlp, c, l, w, ocfg)
        ref_opt.step()
        ref_losses.append(float(out["loss"]))
        cd = c.cuda()
        perm, slots = dp.plan_batch(octree, cd)
        loss, _, _ = fused_train_step(octree, dec, cd, l.cuda(), w.cuda(), step_options(fx), perm=perm, slots=slots)
        opt.step(zero_grad=True)
        losses.append(float(loss))
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (losses, ref_losses)
    for k, r in enumerate(oct_.hier_features):
        assert rel_err(octree.hier_features[k], r) <= 2e-4
    for p, r in zip(dec.fused_params(), mlp.params()):
        assert rel_err(p, r) <= 2e-4


def test_sorted_sampler_is_sorted_uniform_and_reproducible():
    """shine_sample_sorted: ascending indices in range, uniform over the pool (64-bin chi-square), a new stream id gives
    a new draw, the same (seed, stream) reproduces it."""
    from shine_mapping_amd import _lib
    import ctypes as C

    lib = _lib.lib()
    pool, n = 1_000_003, 1 << 16
    need = C.c_size_t(0)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.shine_sample_sorted(pool, n, 7, 0, None, None, 0, None, None, None, C.byref(need), st))
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    draws = []
    for stream_id in (0, 1, 0):
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        _lib.check(lib.shine_sample_sorted(pool, n, 7, stream_id, idx.data_ptr(), None, 0, None, None, ws.data_ptr(), C.byref(need), st))
        torch.cuda.synchronize()
        draws.append(idx.cpu().long())
    a, b, a2 = draws
    assert torch.equal(a, a2) and not torch.equal(a, b)
    for d in (a, b):
        assert bool((d[1:] >= d[:-1]).all()) and int(d.min()) >= 0 and int(d.max()) < pool
        hist = torch.bincount((d * 64) // pool, minlength=64).double()
        chi2 = float(((hist - n / 64) ** 2 / (n / 64)).sum())
        assert chi2 < 130.0, chi2  # 63 dof: mean 63, sd ~11
    # gaps of sorted iid uniforms are ~geometric: about n*(1-exp(-n/pool)) / ... duplicates are expected and allowed
    assert int((a[1:] == a[:-1]).sum()) > 0


@pytest.mark.parametrize("name", ["maicity_bce_L4", "kitti_eik_L3"])
def test_pool_mode_step_equals_batch_mode_on_the_drawn_batch(name):
    """SortedPool: the fused step reading straight out of the node-ordered pool (sorted indices, pool slots) must equal
    the step on the gathered batch, and the oracle on the same points."""
    from oracle import shine_oracle as so
    from shine_mapping_amd import fused_train_step
    from shine_mapping_amd.sampler import SortedPool

    fx = load_golden(name)
    cfg, octree, dec = product_from_golden(fx)
    octree._require_tables(with_ranks=True)
    sp = SortedPool(octree, fx["coord"].cuda(), fx["sdf_label"].cuda(), fx["weight"].cuda(), seed=3)
    idx = sp.draw(1500)
    params = list(octree.hier_features) + dec.fused_params()
    loss_p, pred_p, g_p = fused_train_step(octree, dec, None, None, None, step_options(fx), want_grad_x=True,
                                           pool=sp, idx=idx)
    grads_p = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    c, l, w = sp.get_batch(idx)
    loss_b, pred_b, g_b = fused_train_step(octree, dec, c.contiguous(), l.contiguous(), w.contiguous(),
                                           step_options(fx), want_grad_x=True)
    torch.cuda.synchronize()
    assert abs(float(loss_p) - float(loss_b)) <= 1e-6 * max(1.0, abs(float(loss_b)))
    assert abs_err(pred_p, pred_b) <= 1e-5
    if g_b is not None:
        assert g_close(g_p, g_b, 1e-5)
    for a, b in zip(grads_p, [p.grad for p in params]):
        # (the last decoder tensor, d loss / d b3 = sum of delta over the batch, is ONE heavily cancelling sum — 1.5e-5 from
        # terms of 7e-4: two kernels' summation orders differ by ~1e-9 absolute, which is 1e-4 of the result itself)
        assert rel_err(a, b) <= 2e-5 or abs_err(a, b) <= 1e-7
    ocfg, oct_, mlp = oracle_from_golden(fx)
    ref = so.train_step(oct_, mlp, c.cpu(), l.cpu(), w.cpu(), ocfg)
    assert abs_err(pred_p, ref["pred"]) <= TOL
    for k, r in enumerate(ref["feat_grads"]):
        assert rel_err(grads_p[k], r) <= TOL


def test_incremental_loop_end_to_end():
    """shine_incre.py:86-195 on the fused path: per frame update() -> re-plan the pool -> iterations of
    {sorted draw, fused step (sum reduction) + regulariser, fused Adam} -> importance sweep.  Checks the plumbing
    that the piecewise parity tests do not: table growth + rehash between frames, pool re-planning, optimiser
    re-creation (feature_octree.py:156 re-creates the Parameters), the regulariser actually restraining drift."""
    from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, fused_train_step, synth
    from shine_mapping_amd.incre_learning import cal_feature_importance
    from shine_mapping_amd.ops import fused_regularization, touched_flags
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    cfg = synth.make_config("ncd", device="cuda", lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0)
    torch.manual_seed(0)
    octree, dec = FeatureOctree(cfg), Decoder(cfg)
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum")
    frames = list(synth.make_frames(cfg, frames=3, beams=16, azimuths=120, seed=4, device="cuda"))
    first_loss = last_loss = None
    for fi, (coord, label, weight) in enumerate(frames):
        octree.update(coord[weight > 0], incremental_on=True)
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())  # shine_incre.py:108-109
        octree._require_tables(with_ranks=True)
        pool = SortedPool(octree, coord, label, weight, seed=fi)
        touched = touched_flags(octree)
        for it in range(15):
            idx = pool.draw(2048)
            loss, pred, _ = fused_train_step(octree, dec, None, None, None, opts, pool=pool, idx=idx, touched=touched)
            reg = fused_regularization(octree, cfg.lambda_forget, touched)
            opt.step(zero_grad=True)
            total = float(loss) + cfg.lambda_forget * float(reg)
            assert total == total and abs(total) < 1e12  # finite
            if fi == 0 and it == 0:
                first_loss = float(loss)
            last_loss = float(loss)
        data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
        cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, 2048, 2, "sum")
        assert all(float(t.abs().sum()) > 0 for t in octree.importance_weight)
        assert all(float(t[-1].abs().max()) == 0 for t in octree.importance_weight)
        if fi > 0:
            assert octree._reg_grad_on == [False] * cfg.tree_level_feat  # attached-clone quirk from frame 2 on
    assert last_loss < first_loss  # it learns
    # the octree kept answering exactly like the reference tables would: indices of a probe batch are consistent
    idx_lists = octree.get_indices(frames[-1][0][:512])
    tab = octree.nodes_lookup_tables
    from oracle import kaolin_shim as kal
    for i, ix in enumerate(idx_lists):
        lvl = octree.max_level - i
        codes = kal.points_to_morton(kal.quantize_points(frames[-1][0][:512].cpu(), lvl)).tolist()
        want = torch.tensor([tab[lvl].get(m, [-1] * 8) for m in codes])
        assert torch.equal(ix.cpu(), want)


def test_pipelined_incremental_frames_equal_the_sequential_loop():
    """FeatureOctree.enable_async_growth(): update() grows the tree on a stream of its own, so a loop without any synchronisation
    between frames binds frame k + 1 while the device still trains frame k (bench.py's ncd-incre leg).  The same four frames
    {update, new Adam, pool plan, K graphed iterations with the regulariser, importance sweep}
      * sequentially, with a synchronisation after every frame,
      * pipelined: growth stream, two alternating iteration graphs, no synchronisation, and ~20 ms of unrelated work queued on
        the main stream in front of every update so that the growth certainly runs while earlier work is still pending,
    must end in the same tables, features, decoder, importance and last-frame copies (the training runs in the deterministic
    accumulation mode; the sweep's atomics are unordered: 1e-5)."""
    from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, synth
    from shine_mapping_amd.incre_learning import cal_feature_importance
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    K, N, BS = 6, 1024, 1024
    cfg = synth.make_config("ncd", device="cuda", lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0)
    frames = list(synth.make_frames(cfg, frames=4, beams=16, azimuths=120, seed=9, device="cuda"))
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum", deterministic=True)

    def run(pipelined):
        torch.manual_seed(0)
        octree, dec = FeatureOctree(cfg), Decoder(cfg).cuda()
        if pipelined:
            octree.enable_async_growth()
        busy = torch.randn(2048, 2048, device="cuda")
        keep, caps = [], []
        for fi, (coord, label, weight) in enumerate(frames):
            if pipelined:
                for _ in range(30):  # the main stream stays busy while update() returns
                    busy = torch.tanh(busy @ busy * 1e-3)
                with torch.cuda.stream(octree.growth_stream):
                    surf = coord[weight > 0]
            else:
                surf = coord[weight > 0]
            # ready: the points were selected on the growth's own stream (frames 0, 2, ...: stated; frame 1: an event says so)
            ev = None
            if pipelined and fi % 2 == 1:
                ev = torch.cuda.Event()
                ev.record(octree.growth_stream)
            octree.update(surf, incremental_on=True, ready=(ev if ev is not None else False) if pipelined else None)
            opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
            pool = SortedPool(octree, coord, label, weight, seed=fi, canonical=True)
            it = GraphedIteration(octree, dec, pool, opt, opts, N, lambda_forget=cfg.lambda_forget, unroll=2, eager_first=False,
                                  graph_slot=fi % 2 if pipelined else 0)
            it.run(K)
            data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
            cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, BS, 2, "sum", pool=pool)
            keep.append((opt, pool, it))  # (a pipelined loop's objects outlive the frame in bench.py too: one frame)
            keep = keep[-2:]
            caps.append([octree._tables.stats(s)[0] for s in range(octree.featured_level_num)])
            if not pipelined:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        octree._sync_host()
        return dict(feat=[p.detach().clone() for p in octree.hier_features], dec=[p.detach().clone() for p in dec.fused_params()],
                    imp=[t.clone() for t in octree.importance_weight], last=[t.detach().clone() for t in octree.features_last_frame],
                    keys=[k.copy() for k in octree._node_keys], ids=[k.copy() for k in octree._node_ids], loss=float(it.loss),
                    caps=caps)

    a, b = run(False), run(True)
    # the case that matters most: a hash table was re-built (new arrays, the old ones retired) by a growth that ran while the
    # previous frame's iterations — bound to the old arrays — were still queued
    assert any(later != b["caps"][0] for later in b["caps"][1:]), b["caps"]
    for x, y in zip(a["keys"] + a["ids"], b["keys"] + b["ids"]):
        assert np.array_equal(x, y)
    for key in ("feat", "dec", "imp", "last"):
        for x, y in zip(a[key], b[key]):
            assert x.shape == y.shape and rel_err(y, x) <= 1e-5, key
    assert abs(a["loss"] - b["loss"]) <= 1e-5 * abs(a["loss"])


def _incremental_trajectory(K, N, BS, n_frames, beams, azimuths, freeze_after=None, tier="B", noise_probe=False):
    """Config 4 across frames (shine_incre.py:100-195), product vs CPU oracle on the SAME drawn batches and the SAME fresh
    feature rows: per frame  [freeze_model(geo_mlp) at frame `freeze_after`, :93-97] -> update(incremental_on=True) -> a new Adam
    (:107-109) -> K iterations of {BCE(sum) + lambda_forget * cal_regularization, backward, Adam} -> cal_feature_importance.
    tier "B": the path bench.py's ncd-incre leg runs — device octree growth, SortedPool, loop.GraphedIteration(fold=True,
    eager_first=False, unroll=2) in the deterministic accumulation mode, the fused importance sweep re-using the pool plan.
    tier "A": the UNCHANGED driver's loop body (shine_incre.py:118-181, :185-188) on what `import shine_mapping_amd.dropin` binds
    its names to: query_feature -> sdf (one fused node) -> sdf_bce_loss -> octree.cal_regularization() (OctreeRegularizer) ->
    opt.zero_grad / backward / opt.step (FusedAdam), then cal_feature_importance — fed the batches Tier B's sampler draws.
    After EVERY frame the feature tables, the decoder, importance_weight and features_last_frame must agree.

    Two oracles run side by side (tests/incre_trajectory.py):
      * clean   — the regulariser's gradient as it is in exact arithmetic (live while features_last_frame is the detached
        first-frame copy, model/feature_octree.py:146; zero once it is the attached clone, :160): the product is held to it at
        2e-4 of max-abs (what the batch-mode trajectory test uses; two clean CPU runs that only differ in the summation order of
        a batch are 5e-5 apart after three frames) up to a handful of noise-amplified elements (see the comment at the assert);
      * literal — so.train_step(regularize=True), bit-identical to the reference: from the second frame on autograd adds and
        subtracts 2 lambda imp (F - F_last) in fp32 and leaves rounding noise that Adam (eps 1e-15) amplifies to 5-10 % of
        max-abs on ~1 % of the elements — literal vs clean, both on the CPU, tests/test_oracle.py pins that.  No implementation
        with another rounding can follow THOSE elements; the product is held to the literal oracle in the first frame (where
        the importance is still zero) at 2e-4, and afterwards at the literal oracle's own distance from the clean one.
    noise_probe: a THIRD oracle — the clean one fed the same batches with their points permuted AND every coordinate moved to
    the next representable float (one ulp: ~1e-7 relative on every interpolation weight, the size of the differences any
    implementation with its own exp / log / reciprocal and its own summation order has per sample) — measures how many elements
    the oracle ITSELF moves beyond 2e-4 under a perturbation of that size; the allowance of a tensor is then max(16, 5e-4 of
    it, 3 x that count).  (Permuting alone moves the oracle by 1e-6: measured, profiles/r05_ab_experiments.txt block 5.  At
    K = 10 two such runs stay within 5e-5; at config 4's 50 iterations per frame Adam — eps 1e-15, a normalised step — has five
    times as many steps to turn the relative error of a nearly cancelling coarse-level gradient into a move of that fraction
    of lr per step.)
    -> the allowance actually used: the largest count of elements beyond 2e-4 of max-abs (and the largest deviation) seen."""
    from incre_trajectory import OracleIncremental, deviation
    from oracle import shine_oracle as so
    from shine_mapping_amd import Decoder, FeatureOctree, StepOptions, autograd_ops, sdf_bce_loss, synth
    from shine_mapping_amd.incre_learning import cal_feature_importance
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    cfg = synth.make_config("ncd", device="cuda", lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0)
    torch.manual_seed(0)
    octree, dec = FeatureOctree(cfg), Decoder(cfg).cuda()
    ocfg = so.make_config(tree_level_world=cfg.tree_level_world, tree_level_feat=cfg.tree_level_feat,
                          leaf_vox_size=cfg.leaf_vox_size, sigma_sigmoid_m=cfg.sigma_sigmoid_m, poly_int_on=cfg.poly_int_on,
                          loss_reduction="sum", lambda_forget=cfg.lambda_forget)
    dec_state = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    oracles = {"clean": OracleIncremental(ocfg, lr=cfg.lr, weight_decay=cfg.weight_decay, literal=False, decoder_state=dec_state),
               "literal": OracleIncremental(ocfg, lr=cfg.lr, weight_decay=cfg.weight_decay, literal=True, decoder_state=dec_state)}
    if noise_probe:
        oracles["shuffled"] = OracleIncremental(ocfg, lr=cfg.lr, weight_decay=cfg.weight_decay, literal=False,
                                                decoder_state=dec_state)
    shuffle = torch.randperm(N, generator=torch.Generator().manual_seed(9))
    opts = StepOptions(sigma=cfg.sigma_sigmoid, loss_reduction="sum", deterministic=True)
    frames = list(synth.make_frames(cfg, frames=n_frames, beams=beams, azimuths=azimuths, seed=4, device="cuda"))
    seen_quirk = False
    used = {"elements": 0, "deviation": 0.0}
    torch.set_num_threads(min(8, torch.get_num_threads()))
    for fi, (coord, label, weight) in enumerate(frames):
        if freeze_after is not None and fi == freeze_after:  # shine_incre.py:93-97
            for p in dec.parameters():
                p.requires_grad_(False)
            opts.decoder_grad_on = False
            for o in oracles.values():
                o.freeze_decoder()
        octree.update(coord[weight > 0], incremental_on=True)
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())  # shine_incre.py:107-109
        octree._require_tables(with_ranks=True)
        pool = SortedPool(octree, coord, label, weight, seed=fi, canonical=True)
        # the K batches the graph's device-side sampler will draw (stream ids 0 .. K-1), read back through eager draws
        batches = [pool.get_batch(pool.draw(N)) for _ in range(K)]
        pool.draws, pool._stream_state = 0, None
        rows = [p.detach().cpu() for p in octree.hier_features]
        grew = None
        for name, o in oracles.items():
            grew = o.begin_frame(coord[weight > 0].cpu(), new_rows=rows)
            for c, l, w in batches:
                c, l, w = c.cpu(), l.cpu(), w.cpu()
                if name == "shuffled":
                    c, l, w = torch.nextafter(c[shuffle], torch.full_like(c, float("inf"))), l[shuffle], w[shuffle]
                o.iterate(c, l, w)
            o.end_frame(coord.cpu(), label.cpu(), BS, 2)
        seen_quirk = seen_quirk or (fi > 0 and not all(grew))
        data = type("Pool", (), {"coord_pool": coord, "sdf_label_pool": label})()
        if tier == "B":
            it = GraphedIteration(octree, dec, pool, opt, opts, N, lambda_forget=cfg.lambda_forget, unroll=2, fold=True,
                                  eager_first=False)
            it.run(K - (1 if it.ran_eager else 0))
            total = float(it.loss) + cfg.lambda_forget * float(it.reg)
            cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, BS, 2, "sum", pool=pool)
        else:
            autograd_ops.DETERMINISTIC_BACKWARD = True
            try:
                for c, l, w in batches:  # shine_incre.py:118-181, the names as dropin binds them
                    feature = octree.query_feature(c)
                    sdf_pred = dec.sdf(feature)
                    w_abs = torch.abs(w)
                    cur_loss = 0.
                    cur_loss += sdf_bce_loss(sdf_pred, l, cfg.sigma_sigmoid, w_abs, False, "sum")
                    reg_loss = octree.cal_regularization()
                    cur_loss += cfg.lambda_forget * reg_loss
                    opt.zero_grad(set_to_none=True)
                    cur_loss.backward()
                    opt.step()
                total = float(cur_loss)
                opt.zero_grad(set_to_none=True)  # shine_incre.py:186
                cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, BS, 2, "sum")
            finally:
                autograd_ops.DETERMINISTIC_BACKWARD = False
        torch.cuda.synchronize()
        assert opt.steps_taken() == K
        assert octree._reg_grad_on == oracles["clean"].grad_on
        mine = dict(features=list(octree.hier_features), decoder=dec.fused_params(), importance=octree.importance_weight,
                    features_last=octree.features_last_frame)
        clean, literal = oracles["clean"].state(), oracles["literal"].state()
        noise = oracles["shuffled"].state() if noise_probe else None
        report, bad = [], []
        for key, tensors in mine.items():
            for k, t in enumerate(tensors):
                assert t.shape == clean[key][k].shape
                d_clean, n_clean = deviation(t, clean[key][k], 2e-4)
                d_lit, n_lit = deviation(t, literal[key][k], 2e-4)
                own, n_own = deviation(literal[key][k], clean[key][k], 2e-4)
                report.append((fi, key, k, t.numel(), "%.2e" % d_clean, n_clean, "%.2e" % d_lit, n_lit, "%.2e" % own, n_own))
                # Adam with eps = 1e-15 is a normalised step: it turns the RELATIVE error of an element's gradient into a step
                # error of that fraction of lr.  The per-step gradients agree to ~2e-6 of a tensor's max-abs (asserted by the
                # single-step tests), which is percents of a gradient 1e-4 times smaller than the largest — a coarse-level row
                # whose contributions cancel: measured on the GPU, ONE such row (8 elements) is 4-9e-4 of max-abs away after
                # the first frame, everything else <= 8e-5.  So: at most 16 elements (or 5e-4 of a tensor) may exceed 2e-4 of
                # max-abs, none by more than 1.5 lr per frame (one whole sign flip); a systematic error — a wrong batch, a
                # stale flag, a missing regulariser gradient, a wrong bias correction — moves hundreds of elements by lr.
                allowed = max(16, int(5e-4 * t.numel()))
                if noise is not None:
                    d_noise, n_noise = deviation(noise[key][k], clean[key][k], 2e-4)
                    allowed = max(allowed, 3 * n_noise)
                    report[-1] = report[-1] + ("oracle vs itself reordered: %.2e" % d_noise, n_noise)
                scale = max(float(clean[key][k].abs().max()), 1e-30)
                if n_clean > allowed or (key != "importance" and d_clean * scale > 1.5 * cfg.lr * (fi + 1)):
                    bad.append(("clean", key, k))
                if n_lit > allowed + n_own or (fi == 0 and n_lit > allowed):
                    bad.append(("literal", key, k))
                if n_clean > used["elements"]:
                    used.update(elements=n_clean, of=t.numel(), tensor=(fi, key, k))
                used["deviation"] = max(used["deviation"], d_clean)
        assert not bad, "\n".join(str(r) for r in [bad] + report)
        ref_clean, ref_lit = oracles["clean"].losses[-1], oracles["literal"].losses[-1]
        assert abs(total - ref_clean) <= 2e-4 * max(1.0, abs(ref_clean)), (fi, "clean", total, ref_clean)
        # the literal oracle: at its own distance from the clean one (its noise-amplified features enter lambda_forget = 1e4 times
        # the regulariser: 1 % of the loss after 10 iterations per frame, 5 % after 50), exactly in the first frame
        lit_own = 0.0 if fi == 0 else 1.5 * abs(ref_lit - ref_clean)
        assert abs(total - ref_lit) <= 2e-4 * max(1.0, abs(ref_lit)) + lit_own, (fi, "literal", total, ref_lit, ref_clean)
    assert all(not g for g in octree._reg_grad_on) or seen_quirk  # every level grew again: the attached-clone quirk is live
    if freeze_after is not None:
        assert all(not p.requires_grad for p in dec.parameters()) and n_frames > freeze_after
    print("incremental trajectory tier %s K=%d N=%d frames=%d: allowance used — %s" % (tier, K, N, n_frames, used))
    return used


def test_incremental_trajectory_matches_oracle():
    """Three frames x 10 iterations at N = 1024 through Tier B (round 4's form of the test)."""
    _incremental_trajectory(K=10, N=1024, BS=1024, n_frames=3, beams=16, azimuths=120)


def test_incremental_trajectory_at_config_4_shape_with_the_decoder_frozen_mid_run():
    """config 4's real shape — N = 4096, 50 iterations per frame (config/ncd/ncd_incre_reg.yaml:54,52) —
    over four frames with freeze_after_frame crossing INSIDE the run (the decoder trains in frames 0-1 and is frozen from frame 2
    on, shine_incre.py:93-97): the frozen launches (decoder_grad_on = 0), the optimiser re-created without the decoder's
    gradients, the graph rebuilt for another kernel instantiation.  Same clean / literal oracle pair, same allowance rule; the
    allowance actually used is printed (pytest -s) so that a regression shows before it fails."""
    used = _incremental_trajectory(K=50, N=4096, BS=4096, n_frames=4, beams=32, azimuths=240, freeze_after=2, noise_probe=True)
    assert used["deviation"] < 0.5  # (of max-abs: one sign flip of one element is ~lr / max-abs)


def test_tier_a_incremental_frames_match_the_oracle_trajectory():
    """config 4 through the UNCHANGED driver's names — two frames of the shine_incre.py loop body on the
    drop-in classes (the regulariser as autograd_ops.OctreeRegularizer, the sweep as shine_importance_sweep) against the same
    oracle trajectory Tier B is held to."""
    _incremental_trajectory(K=10, N=1024, BS=1024, n_frames=3, beams=16, azimuths=120, tier="A")


def test_retired_table_arrays_are_accounted_and_trimmed():
    """device arrays a growth replaces (a rehashed hash table, an outgrown scratch buffer) are retired inside
    the table handle, not freed on the spot; FeatureOctree.retired_table_bytes() reports them and trim_tables() frees them —
    after which the tables still answer queries as before.  The importance sweep's scratch is given back by release_scratch()."""
    from shine_mapping_amd import Decoder, FeatureOctree, incre_learning, synth

    cfg = synth.make_config("ncd", device="cuda")
    frames = list(synth.make_frames(cfg, frames=6, beams=32, azimuths=240, seed=7, device="cuda"))
    torch.manual_seed(0)
    octree, dec = FeatureOctree(cfg), Decoder(cfg).cuda()
    for coord, label, weight in frames:
        octree.update(coord[weight > 0], incremental_on=True)
    coord = frames[-1][0][:5000].contiguous()
    with torch.no_grad():
        before = octree.query_feature(coord).clone()
        idx_before = [t.clone() for t in octree.get_indices(coord)]
    held = octree.retired_table_bytes()
    assert held > 0  # six frames of growth rehashed the tables at least once
    assert octree.trim_tables() == held and octree.retired_table_bytes() == 0
    with torch.no_grad():
        assert torch.equal(octree.query_feature(coord), before)
    assert all(torch.equal(a, b) for a, b in zip(octree.get_indices(coord), idx_before))
    octree.trim_retired_above = 0  # the automatic form: update() trims whatever its growth retired
    more = list(synth.make_frames(cfg, frames=12, beams=32, azimuths=240, seed=8, device="cuda"))[6:]
    for c, l, w in more:
        octree.update(c[w > 0], incremental_on=True)
    assert octree.retired_table_bytes() == 0
    data = type("Pool", (), {"coord_pool": frames[-1][0], "sdf_label_pool": frames[-1][1]})()
    incre_learning.cal_feature_importance(data, octree, dec, cfg.sigma_sigmoid, 4096, 2, "sum")
    torch.cuda.synchronize()
    assert len(incre_learning._SCRATCH) == 1
    incre_learning.release_scratch()
    assert len(incre_learning._SCRATCH) == 0 and len(incre_learning._BUDGET) == 0


def test_stale_pool_is_rejected_after_octree_growth():
    from shine_mapping_amd import StepOptions, fused_train_step
    from shine_mapping_amd.sampler import SortedPool

    fx = load_golden("maicity_bce_L3")
    cfg, octree, dec = product_from_golden(fx)
    sp = SortedPool(octree, fx["coord"].cuda(), fx["sdf_label"].cuda(), fx["weight"].cuda())
    idx = sp.draw(256)
    fused_train_step(octree, dec, None, None, None, step_options(fx), pool=sp, idx=idx)  # fine
    octree.update(torch.tensor([[0.31, 0.27, -0.11], [0.33, 0.27, -0.11]]))              # new nodes -> slots may move
    with pytest.raises(RuntimeError, match="rebuild"):
        fused_train_step(octree, dec, None, None, None, step_options(fx), pool=sp, idx=idx)
    sp.rebuild(fx["coord"].cuda(), fx["sdf_label"].cuda(), fx["weight"].cuda())
    fused_train_step(octree, dec, None, None, None, step_options(fx), pool=sp, idx=idx)


def test_batch_training_learns_the_synthetic_map():
    """Functional end-to-end check (the reference validates by meshing; here: the implicit map must fit its data).
    300 fused iterations (sorted pool draws, fused Adam) on a small MaiCity-like scene: the loss on fresh pool samples
    drops well below its initial value and sign(pred) agrees with the sign of the sampled projective SDF label."""
    from shine_mapping_amd import StepOptions, forward_sdf, fused_train_step, synth
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    wl = synth.build_workload("maicity", frames=8, device="cuda", seed=11, tree_level_feat=3, azimuths=300, beams=32,
                              lr=0.01, opt_adam=True, adam_eps=1e-15, lr_level_reduce_ratio=1.0)
    cfg, octree, dec = wl.cfg, wl.octree, wl.decoder
    opts = StepOptions(sigma=cfg.sigma_sigmoid)
    opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
    octree._require_tables(with_ranks=True)
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=5)
    losses = []
    for it in range(300):
        idx = sp.draw(1 << 15)
        loss, _, _ = fused_train_step(octree, dec, None, None, None, opts, pool=sp, idx=idx)
        opt.step(zero_grad=True)
        if it % 50 == 0 or it == 299:
            losses.append(float(loss))
    assert losses[-1] < 0.75 * losses[0], losses
    g = torch.Generator(device="cuda").manual_seed(3)
    c, l, w = synth.draw_batch(wl.pool, 1 << 15, g)
    pred = forward_sdf(octree, dec, c)["pred"]
    near = l.abs() > 0.3 * cfg.surface_sample_range_m * cfg.scale  # skip labels too close to zero to have a sign
    agree = ((pred > 0) == (l > 0))[near].float().mean()
    assert float(agree) > 0.85, float(agree)


# ---------------------------------------------------------------------------------------------------------
# meshing query: Mesher.get_query_from_bbx + query_points, utils/mesher.py:33-152
class _Box:
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def get_min_bound(self):
        import numpy as np
        return np.asarray(self.lo, dtype=np.float64)

    def get_max_bound(self):
        import numpy as np
        return np.asarray(self.hi, dtype=np.float64)


@pytest.mark.parametrize("name", ["mesh_query_L3", "mesh_query_L4"])
def test_mesher_query_points_matches_reference(name):
    import numpy as np
    from oracle import shine_oracle as so
    from shine_mapping_amd.mesher import Mesher

    fx = load_golden(name)
    cfg, octree, dec = product_from_golden(load_golden(fx["source"]))
    cfg.mc_vis_level = fx["mc_vis_level"]
    cfg.pad_voxel = fx["pad_voxel"]
    mesher = Mesher(cfg, octree, dec.cuda(), None)
    coord, num, origin = mesher.get_query_from_bbx(_Box(fx["lo"], fx["hi"]), fx["voxel"])
    ref_coord, ref_num, ref_origin = so.grid_query_coords(fx["lo"], fx["hi"], fx["voxel"], cfg.scale, fx["pad_voxel"])
    assert coord.is_cuda and torch.equal(coord.cpu(), ref_coord), "grid coordinates must be bit-identical"
    assert np.array_equal(num, ref_num) and np.array_equal(origin, ref_origin)
    sdf, sem, mask = mesher.query_points(coord, fx["bs"], True, False, True)
    assert sem is None
    assert str(sdf.dtype) == fx["sdf_dtype"]  # float64 buffers when chunked, float32 otherwise (mesher.py:43-53,90-104)
    assert np.array_equal(mask.astype("uint8"), fx["mc_mask"].numpy()), "marching-cubes mask must be exact"
    assert np.abs(sdf.astype("float32") - fx["sdf_pred"].numpy()).max() <= TOL
    # options: mask only / sdf only
    s2, _, m2 = mesher.query_points(coord, 10 ** 9, True, False, False)
    assert m2 is None and np.abs(s2 - fx["sdf_pred"].numpy()).max() <= TOL
    s3, _, m3 = mesher.query_points(coord, 10 ** 9, False, False, True)
    assert s3 is None and np.array_equal(m3.astype("uint8"), fx["mc_mask"].numpy())


def test_mesher_query_matches_forward_at_scale():
    """2^21 grid points: the meshing kernel against the step kernel's forward (same tables, different code)."""
    from shine_mapping_amd import forward_sdf, synth
    from shine_mapping_amd.mesher import query_points_device

    wl = synth.build_workload("maicity", frames=8, device="cuda", seed=5, tree_level_feat=4, azimuths=300)
    octree, dec = wl.octree, wl.decoder.cuda()
    with torch.no_grad():
        for p in octree.hier_features:
            p.mul_(6.0)
    g = torch.Generator(device="cuda").manual_seed(3)
    lo, hi = wl.pool.coord.min(0).values, wl.pool.coord.max(0).values
    coord = lo + (hi - lo) * torch.rand((1 << 21, 3), device="cuda", generator=g)
    out = forward_sdf(octree, dec, coord, want_indices=True)
    for lvl in range(4):
        sdf, mask = query_points_device(octree, dec, coord, check_level=lvl)
        assert torch.equal(mask, (out["indices"][lvl] >= 0).all(1))
        assert (sdf + out["pred"]).abs().max().item() <= TOL


# ---------------------------------------------------------------------------------------------------------
# GPU octree growth: FeatureOctree.update on CUDA points = shine_tables_grow
@pytest.mark.parametrize("name", ["maicity_bce_L3", "maicity_bce_L4", "kitti_eik_L3", "ncd_reg_L3", "linear_L2_nopoly"])
def test_device_octree_build_matches_reference_tables(name):
    """The frames the real reference built its tables from (fixture surface_points), grown on the device: the same
    node -> corner-id tables, in the same insertion order, and queries through them hit the reference's indices."""
    from shine_mapping_amd import FeatureOctree, synth

    fx = load_golden(name)
    cfg = synth.make_config("maicity", device="cuda", **fx["cfg"])
    octree = FeatureOctree(cfg)
    for sp in fx["surface_points"]:
        octree.update(sp.cuda(), incremental_on=bool(fx["regularize"]))
    assert octree._dev_frames, "CUDA points must take the device path"
    idx = octree.get_indices(fx["coord"].cuda())  # straight through the device tables, before any host sync
    for k, r in enumerate(fx["out"]["indices"]):
        assert torch.equal(idx[k].cpu(), r)
    for s, (keys, ids) in enumerate(fx["tables"]):
        lvl = octree.free_level_num + s
        mine = octree.nodes_lookup_tables[lvl]
        assert list(mine.keys()) == keys.tolist(), "level %d insertion order differs" % lvl
        assert mine == dict(zip(keys.tolist(), ids.tolist())), "level %d tables differ" % lvl
        assert octree.hier_features[s].shape == fx["features"][s].shape
        assert torch.count_nonzero(octree.hier_features[s][-1]) == 0
    if fx["regularize"]:
        assert octree._reg_grad_on == [False] * cfg.tree_level_feat


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_device_update_equals_oracle_update_over_random_frame_sequences(seed):
    """shine_tables_grow vs OracleOctree.update (the reference's loops, pinned to the real reference) over random
    multi-frame sequences: duplicates, points outside the cube, an empty frame, a frame that adds nothing, a
    host-side update in the middle (mixed mode), a pickle round trip."""
    import pickle
    import numpy as np
    from oracle import shine_oracle as so
    from shine_mapping_amd import FeatureOctree, synth

    rng = np.random.default_rng(100 + seed)
    L = int(rng.integers(1, 5))
    world = int(rng.integers(max(L, 5), 10))
    over = dict(tree_level_world=world, tree_level_feat=L, leaf_vox_size=float(rng.choice([0.2, 0.5, 1.0])))
    ocfg = so.make_config(**over)
    cfg = synth.make_config("maicity", device="cuda", **over)
    mine, ref = FeatureOctree(cfg), so.OracleOctree(ocfg)
    incremental = bool(rng.integers(0, 2))
    centre = rng.uniform(-0.5, 0.5, 3)
    prev = None
    for frame in range(7):
        n = int(rng.integers(1, 3000))
        spread = float(rng.choice([0.01, 0.1, 0.4]))
        if frame == 3 and prev is not None:  # adds nothing new
            pts = prev
        elif frame == 4:
            pts = torch.zeros((0, 3))
        else:
            pts = torch.from_numpy((centre + rng.normal(0, spread, (n, 3))).astype(np.float32)).clamp(-1.2, 1.2)
            pts = torch.cat((pts, pts[: n // 3]))  # duplicates
        prev = pts
        on_host = frame == 5 and seed % 2 == 0  # mixed mode: one host-side update between device-side ones
        torch.manual_seed(1000 + frame)
        if pts.shape[0]:
            mine.update(pts if on_host else pts.cuda(), incremental)
        torch.manual_seed(1000 + frame)
        if pts.shape[0]:  # (the oracle draws its rows from the CPU generator: shapes and tables are compared)
            ref.update(pts, incremental)
        if frame == 2 and seed % 3 == 0:
            mine = pickle.loads(pickle.dumps(mine))
        for s in range(L):
            lvl = mine.free_level_num + s
            assert mine.nodes_lookup_tables[lvl] == ref.node_table[lvl], (frame, lvl)
            assert list(mine.nodes_lookup_tables[lvl].keys()) == list(ref.node_table[lvl].keys()), (frame, lvl)
            assert mine.corners_lookup_tables[lvl] == ref.corner_table[lvl], (frame, lvl)
            if len(ref.hier_features) > s:
                assert mine.hier_features[s].shape == ref.hier_features[s].shape
                assert torch.count_nonzero(mine.hier_features[s][-1]) == 0
                if incremental:
                    assert mine.features_last_frame[s].shape == ref.features_last_frame[s].shape
                    assert mine.importance_weight[s].shape == ref.importance_weight[s].shape
                    assert mine.features_last_frame[s].requires_grad == ref.features_last_frame[s].requires_grad
                    assert mine._reg_grad_on[s] == (not ref.features_last_frame[s].requires_grad)
    # the grown tables serve queries: indices against the oracle's dict lookups
    q = torch.from_numpy((centre + rng.normal(0, 0.2, (4000, 3))).astype(np.float32))
    mine_idx = mine.get_indices(q.cuda())
    ref_idx = ref.get_indices(q)
    for a, b in zip(mine_idx, ref_idx):
        assert torch.equal(a.cpu(), b)


def test_device_update_draws_the_same_rows_as_host_update():
    """Same torch seed, same frames: device-side growth appends bit-identical feature rows (same randn calls in the
    same order) and reports the same sort box as the host path."""
    from shine_mapping_amd import FeatureOctree, synth

    fx = load_golden("ncd_reg_L3")
    cfg = synth.make_config("maicity", device="cuda", **fx["cfg"])
    a, b = FeatureOctree(cfg), FeatureOctree(cfg)
    for i, sp in enumerate(fx["surface_points"]):
        torch.manual_seed(7 + i)
        a.update(sp, True)          # host path (CPU points), rows drawn on cuda
        torch.manual_seed(7 + i)
        b.update(sp.cuda(), True)   # device path
    for s in range(cfg.tree_level_feat):
        assert torch.equal(a.hier_features[s], b.hier_features[s])
        assert torch.equal(a.features_last_frame[s], b.features_last_frame[s])
    assert a._sort_box() == b._sort_box()


def test_device_node_ranks_equal_host_ranks():
    """shine_tables_rank_nodes (device sort) against FeatureOctree._host_node_ranks uploaded with
    shine_tables_set_ranks: the planned order of a batch must not change."""
    import ctypes as C
    from shine_mapping_amd import _lib, dp, synth

    wl = synth.build_workload("maicity", frames=5, device="cuda", seed=9, tree_level_feat=4, azimuths=300)
    octree = wl.octree
    coord, _, _ = synth.draw_batch(wl.pool, 1 << 16)
    perm_dev, slots_dev = dp.plan_batch(octree, coord)
    perm_dev, slots_dev = perm_dev.clone(), slots_dev.clone()
    n_buckets = octree._n_buckets
    ranks = octree._host_node_ranks()
    assert n_buckets == sum(r.size for r in ranks) + 64
    lib = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    for s in range(octree.featured_level_num):
        k = torch.from_numpy(octree._node_keys[s]).cuda()
        r = torch.from_numpy(ranks[s]).cuda()
        _lib.check(lib.shine_tables_set_ranks(octree._tables.handle, s, k.data_ptr(), r.data_ptr(), k.numel(),
                                              n_buckets, stream), "shine_tables_set_ranks")
        torch.cuda.synchronize()
    perm_host, slots_host = dp.plan_batch(octree, coord)
    # the order inside a bucket comes from returning atomics (not deterministic); the bucket sequence is: every sorted
    # position must see the same node slots on all levels
    assert torch.equal(slots_dev, slots_host)
    assert torch.equal(perm_dev.sort().values, perm_host.sort().values)
    assert (slots_dev[1:, -1] != slots_dev[:-1, -1]).sum().item() > 100  # slots are in planned order, many runs


# ---------------------------------------------------------------------------------------------------------
# graph-replayable iteration (loop.GraphedIteration): device-resident sampler stream id and Adam step count
@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("mode", ["bce", "incremental", "eikonal"])
def test_graphed_iteration_matches_eager_loop(mode, native):
    """K iterations through ONE HIP graph == K eager iterations: same batches (the stream id advances on the device), same
    Adam bias corrections (the step count advances on the device), same parameters.  native: the graph is built and bound by
    the library (shine_iter_graph_*) / captured from the stream by torch."""
    from shine_mapping_amd import StepOptions, fused_train_step, synth
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.ops import fused_regularization, touched_flags
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    incremental = mode == "incremental"
    K, N = 12, 4096

    def make():
        fx = load_golden("ncd_reg_L3" if incremental else ("kitti_eik_L3" if mode == "eikonal" else "maicity_bce_L3"))
        cfg, octree, dec = product_from_golden(fx)
        dec = dec.cuda()
        cfg.lr, cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio, cfg.weight_decay = 0.01, True, 1e-15, 1.0, 1e-7
        if incremental:
            octree._reg_grad_on = [True] * cfg.tree_level_feat  # exercise the regulariser's gradient path too
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
        octree._require_tables(with_ranks=True)
        # canonical: the samples of a node in pool-index order (the plan leaves them in atomic-retirement order, which differs
        # from one build of the pool to the next — the same indices would then draw different samples in the two loops)
        pool = SortedPool(octree, fx["coord"].cuda().repeat(8, 1), fx["sdf_label"].cuda().repeat(8),
                          fx["weight"].cuda().repeat(8), seed=5, canonical=True)
        # eikonal: the graphed loop takes the surface count from the draw (per-block parts), the eager loop from torch
        opts = StepOptions(sigma=fx["sigma"], loss_reduction="sum" if incremental else "mean", deterministic=True,
                           ekional_loss_on=mode == "eikonal", weight_e=0.1)
        return cfg, octree, dec, opt, pool, opts

    # eager reference loop (host-side stream ids 0..K-1, host-side step count)
    cfg, octree, dec, opt, pool, opts = make()
    touched = touched_flags(octree) if incremental else None
    eager_idx = []
    for it in range(K):
        idx = pool.draw(N)
        eager_idx.append(idx.clone())
        fused_train_step(octree, dec, None, None, None, opts, pool=pool, idx=idx, touched=touched)
        if incremental:
            fused_regularization(octree, 1e3, touched)
        opt.step(zero_grad=True)
    eager_idx.append(pool.draw(N).clone())  # (the batch iteration K + 1 would use)
    pool.draws -= 1
    eager_feats = [p.detach().clone() for p in octree.hier_features]
    eager_mlp = [p.detach().clone() for p in dec.fused_params()]

    # the same K iterations: 1 eager inside the constructor + K-1 replays of the captured graph
    cfg, octree2, dec2, opt2, pool2, opts2 = make()
    step = GraphedIteration(octree2, dec2, pool2, opt2, opts2, N, lambda_forget=1e3 if incremental else 0.0, native=native)
    assert step.native == native
    first = 1 if step.ran_eager else 0  # (the torch-captured form runs iteration 1 eagerly in its constructor)
    # the optimiser's launch draws the NEXT iteration's batch: after iteration k, `_idx` holds batch k + 1
    assert step._ahead and torch.equal(step._idx, eager_idx[first])
    seen = []
    for it in range(first, K):
        loss = step()
        seen.append(step._idx.clone())
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(seen, eager_idx[first + 1:])), "replays must draw the eager loop's batches"
    assert not torch.equal(seen[0], seen[1])
    assert opt2.steps_taken() == K and float(loss) == float(loss)
    # Both loops run the step in its DETERMINISTIC mode (StepOptions.deterministic: one wave walks the batch, the feature-grad
    # atomics are applied in stream order), so graph replay and eager launches must produce the same parameters — not "within
    # the noise of the atomics' order" (Adam's first steps move an element by lr * sign(g), and a near-cancelling gradient sum
    # takes its sign from that order: without the mode two runs of the SAME loop differ by 0.02-0.05 of max-abs).
    for a, b in zip(eager_mlp, dec2.fused_params()):
        assert rel_err(b.detach(), a) <= 1e-6
    for a, b in zip(eager_feats, octree2.hier_features):
        assert rel_err(b.detach(), a) <= 1e-6
    probe = eager_idx[0]
    opts_probe = StepOptions(sigma=opts.sigma, loss_reduction=opts.loss_reduction, deterministic=True,
                             ekional_loss_on=opts.ekional_loss_on, weight_e=opts.weight_e)
    l1, _, _ = fused_train_step(octree, dec, None, None, None, opts_probe, pool=pool, idx=probe)
    l2, _, _ = fused_train_step(octree2, dec2, None, None, None, opts_probe, pool=pool2, idx=probe)
    assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l1))
    # growth invalidates the captured pointers
    octree2.update(torch.tensor([[0.31, 0.27, -0.11], [0.33, 0.27, -0.11]]).cuda())
    with pytest.raises(RuntimeError, match="GraphedIteration"):
        step()


def test_iteration_graph_sees_decoder_weights_written_between_its_launches():
    """Inside the library-built graph the small-batch kernel copies the decoder's MFMA operand image instead of building it
    (V1Args::op_image): the graph's tail nodes keep the image current, and every launch rebuilds it first from the decoder as it
    is then.  So weights written by somebody else between two run() calls — torch, a checkpoint load — must be what the next
    iterations use: same parameters as the eager loop (which builds the image in-kernel every step) that got the same writes."""
    from shine_mapping_amd import StepOptions, fused_train_step
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    N = 4096

    def make():
        fx = load_golden("maicity_bce_L3")
        cfg, octree, dec = product_from_golden(fx)
        dec = dec.cuda()
        cfg.lr, cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio, cfg.weight_decay = 0.01, True, 1e-15, 1.0, 1e-7
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
        octree._require_tables(with_ranks=True)
        pool = SortedPool(octree, fx["coord"].cuda().repeat(8, 1), fx["sdf_label"].cuda().repeat(8),
                          fx["weight"].cuda().repeat(8), seed=5, canonical=True)
        return octree, dec, opt, pool, StepOptions(sigma=fx["sigma"], deterministic=True)

    def perturb(dec, k):
        with torch.no_grad():
            for j, p in enumerate(dec.fused_params()):
                p.mul_(1.0 + 0.01 * (k + 1)).add_(0.003 * (j + 1))

    octree, dec, opt, pool, opts = make()
    for k in range(3):  # eager: 3 x {2 iterations, somebody rewrites the decoder}
        for _ in range(2):
            fused_train_step(octree, dec, None, None, None, opts, pool=pool, idx=pool.draw(N))
            opt.step(zero_grad=True)
        perturb(dec, k)
    want = [p.detach().clone() for p in list(octree.hier_features) + dec.fused_params()]

    octree2, dec2, opt2, pool2, opts2 = make()
    it = GraphedIteration(octree2, dec2, pool2, opt2, opts2, N, unroll=2)
    assert it.native
    for k in range(3):
        it.run(2)
        perturb(dec2, k)
    torch.cuda.synchronize()
    for a, b in zip(want, list(octree2.hier_features) + dec2.fused_params()):
        assert rel_err(b.detach(), a) <= 1e-6


def test_iteration_graph_rebound_while_its_replays_are_still_queued():
    """Two maps share the library's iteration graph (one per device and unroll).  Map A's replays are queued behind a long
    kernel, map B binds the graph straight away — the commit must not rewrite kernel arguments that A's queued replays still
    need (shine_iter_graph_commit waits for the graph's own last replay) — then A again.  Each map ends exactly where the same
    iterations end when run with a synchronisation after every call."""
    from shine_mapping_amd import StepOptions
    from shine_mapping_amd.loop import GraphedIteration
    from shine_mapping_amd.optim import setup_optimizer
    from shine_mapping_amd.sampler import SortedPool

    K, N = 8, 4096

    def make(name, seed):
        fx = load_golden(name)
        cfg, octree, dec = product_from_golden(fx)
        dec = dec.cuda()
        cfg.lr, cfg.opt_adam, cfg.adam_eps, cfg.lr_level_reduce_ratio, cfg.weight_decay = 0.01, True, 1e-15, 1.0, 1e-7
        opt = setup_optimizer(cfg, list(octree.parameters()), dec.fused_params())
        octree._require_tables(with_ranks=True)
        pool = SortedPool(octree, fx["coord"].cuda().repeat(8, 1), fx["sdf_label"].cuda().repeat(8),
                          fx["weight"].cuda().repeat(8), seed=seed, canonical=True)
        opts = StepOptions(sigma=fx["sigma"], deterministic=True)
        return octree, dec, GraphedIteration(octree, dec, pool, opt, opts, N)

    def run(overlapped):
        oa, da, a = make("maicity_bce_L3", 3)
        ob, db, b = make("ncd_reg_L3", 4)
        assert a.native and b.native
        torch.cuda.synchronize()
        if overlapped:
            # ~10 ms of device work in front of A's replays: they are still queued when B (and then A again) binds the graph
            big = torch.randn(4096, 4096, device="cuda")
            for _ in range(40):
                big = big @ big.t() * 1e-4
        for step in (a, b, a, b):
            step.run(K)
            if not overlapped:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [p.detach().clone() for m in ((oa, da), (ob, db)) for p in list(m[0].hier_features) + m[1].fused_params()]

    ref, got = run(False), run(True)
    for x, y in zip(ref, got):
        assert torch.equal(x, y)


@pytest.mark.parametrize("kind,levels,n", [("maicity", 4, (1 << 18) + 37), ("kitti", 3, (1 << 17) + 1)])
def test_pool_mode_at_scale_equals_planned_batch_mode(kind, levels, n):
    """BASELINE-size pool-mode launch (4 tiles per wave, ragged tail, indices prefetched two tiles ahead) against the
    same points pushed through the planned-batch path and through the v0 checker kernel."""
    from shine_mapping_amd import StepOptions, dp, fused_train_step, synth
    from shine_mapping_amd.sampler import SortedPool

    wl = synth.build_workload(kind, frames=8, device="cuda", seed=21, tree_level_feat=levels, azimuths=300)
    octree, dec, cfg = wl.octree, wl.decoder.cuda(), wl.cfg
    with torch.no_grad():
        for p in octree.hier_features:
            p.mul_(5.0)
    octree._require_tables(with_ranks=True)
    sp = SortedPool(octree, wl.pool.coord, wl.pool.sdf_label, wl.pool.weight, seed=2)
    idx = sp.draw(n)
    params = list(octree.hier_features) + dec.fused_params()

    def run(**kw):
        for p in params:
            p.grad = None
        opts = StepOptions(sigma=cfg.sigma_sigmoid, ekional_loss_on=cfg.ekional_loss_on, weight_e=cfg.weight_e,
                           kernel_variant=kw.pop("variant", 0))
        loss, pred, g = fused_train_step(octree, dec, *kw.pop("batch"), opts, want_grad_x=True, **kw)
        torch.cuda.synchronize()
        return float(loss), pred.clone(), None if g is None else g.clone(), [p.grad.clone() for p in params]

    c, l, w = (t.contiguous() for t in sp.get_batch(idx))
    lp, pp, gp, grp = run(batch=(None, None, None), pool=sp, idx=idx)
    perm, slots = dp.plan_batch(octree, c)
    lb, pb, gb, grb = run(batch=(c, l, w), perm=perm, slots=slots)
    l0, p0, g0, gr0 = run(batch=(c, l, w), variant=1)  # v0: lane = point, per-lane atomics
    for other_loss, other_pred, other_g, other_gr in ((lb, pb, gb, grb), (l0, p0, g0, gr0)):
        assert abs(lp - other_loss) <= 2e-5 * max(1.0, abs(other_loss))
        assert abs_err(pp, other_pred) <= 2e-5
        if gp is not None:
            assert g_close(gp, other_g, 5e-5)
        for a, b in zip(grp, other_gr):
            assert rel_err(a, b) <= 1e-4
