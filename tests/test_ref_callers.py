"""The reference's own CALLER code against the drop-ins.

tests/golden/caller_*.npz were recorded by tools/make_golden_callers.py: the reference's ``GaussianHead.forward``
(model/head/gaussian_head.py:122-197) and ``DeformableFeatureAggregation.forward``
(model/encoder/gaussian_encoder/deformable_module.py:146-248), imported unmodified with stub mmengine / mmseg
registries (tests/ref_shim.py), running on this repository's drop-in packages with the raw kernels served by the
CPU oracle.  Here:

* CPU (``-m "not gpu"``, build container only -- skipped where /root/reference is absent): the reference classes
  import against the drop-ins, construct from config-shaped kwargs, expose the reference's state-dict keys, and
  re-running them reproduces the fixture;
* GPU: the same module-level calls the reference caller made are replayed through the drop-in modules on the
  HIP kernels (outputs, labels and every gradient against the fixture), and through the fused entry points
  (``GaussianArgs`` + ``forward_from_rotations``, ``deformable_prepare``).
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEAD_CASES = ["base", "prob", "prob_geosem", "prob_fast"]


def _load(name):
    return np.load(os.path.join(GOLDEN, f"caller_{name}.npz"))


def _cuda_kwargs(d, prob):
    return dict(scale_multiplier=4 if prob else 3, H=int(d["grid_H"]), W=int(d["grid_W"]), D=int(d["grid_D"]),
                pc_min=[float(v) for v in d["grid_pc_min"]], grid_size=float(d["grid_grid_size"]))


# ------------------------------------------------------------------------------------------------ CPU
def test_reference_callers_import_and_construct_on_the_dropins():
    import ref_shim
    if not ref_shim.available():
        pytest.skip("/root/reference absent (GPU box)")
    ref = ref_shim.load_reference()
    import local_aggregate, local_aggregate_prob, local_aggregate_prob_fast
    d = _load("head_base")
    head = ref.GaussianHead(apply_loss_type="random_1", num_classes=18, with_empty=True,
                            empty_args=dict(mean=[0, 0, -1.0], scale=[100, 100, 8.0]), cuda_kwargs=_cuda_kwargs(d, False))
    assert type(head.aggregator) is local_aggregate.LocalAggregator
    assert "aggregator.pc_min" in head.state_dict()                 # eval.py:108 loads checkpoints with strict=True
    head = ref.GaussianHead(apply_loss_type="all", use_localaggprob=True, cuda_kwargs=_cuda_kwargs(d, True))
    assert type(head.aggregator) is local_aggregate_prob.LocalAggregator
    head = ref.GaussianHead(apply_loss_type="fixed_0", use_localaggprob=True, use_localaggprob_fast=True,
                            cuda_kwargs=dict(radii_min=2, **_cuda_kwargs(d, True)))
    assert type(head.aggregator) is local_aggregate_prob_fast.LocalAggregator and head.aggregator.radii_min == 2
    from model.encoder.gaussian_encoder.ops import DeformableAggregationFunction
    assert ref.DAF is DeformableAggregationFunction
    dfa = ref_shim.build_from_cfg(dict(type="DeformableFeatureAggregation", embed_dims=32, num_groups=4, num_levels=3,
                                       num_cams=3, use_deformable_func=True, use_camera_embed=True, residual_mode="cat",
                                       kps_generator=dict(type="SparseGaussian3DKeyPointsGenerator", num_learnable_pts=2,
                                                          fix_scale=[[0, 0, 0]], pc_range=[-1, -1, -1, 1, 1, 1],
                                                          scale_range=[0.1, 0.6])), ref_shim.MODELS)
    assert dfa.use_deformable_func and dfa.num_pts == 3


def test_reference_head_reproduces_the_fixture_on_cpu():
    import torch
    import ref_shim
    if not ref_shim.available():
        pytest.skip("/root/reference absent (GPU box)")
    ref = ref_shim.load_reference()
    d = _load("head_base")
    saved = getattr(torch.Tensor, "cuda")
    torch.Tensor.cuda = lambda self, *a, **k: self          # gaussian_head.py:119 on a CPU-only box
    try:
        head = ref.GaussianHead(apply_loss_type="random_1", num_classes=18, with_empty=True,
                                empty_args=dict(mean=[0, 0, -1.0], scale=[100, 100, 8.0]), cuda_kwargs=_cuda_kwargs(d, False))
        t = lambda k: torch.from_numpy(d[k])
        rep = [dict(gaussian=ref.GaussianPrediction(means=t("means"), scales=t("scales"), rotations=t("rotations"),
                                                    opacities=t("opacities"), semantics=t("semantics")))]
        H, W, D = int(d["grid_H"]), int(d["grid_W"]), int(d["grid_D"])
        metas = dict(occ_xyz=t("occ_xyz"), occ_label=torch.zeros(1, H, W, D, dtype=torch.long),
                     occ_cam_mask=torch.ones(1, H, W, D, dtype=torch.bool))
        with ref_shim.cpu_kernels():
            out = head(rep, metas)
    finally:
        torch.Tensor.cuda = saved
    assert np.allclose(out["pred_occ"][-1].detach().numpy(), d["pred_occ"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(out["final_occ"].numpy(), d["final_occ"])


def test_key_point_restatement_matches_the_reference_caller():
    """oracle/daf_prepare_ref.key_points + project_points reproduce the sampling locations the reference's
    SparseGaussian3DKeyPointsGenerator / project_points handed to the op."""
    import torch
    from oracle import daf_prepare_ref
    d = _load("dfa")
    t = lambda k: torch.from_numpy(d[k])
    kp = daf_prepare_ref.key_points(t("anchor"), t("instance_feature"),
                                    [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]],
                                    t("param_kps_generator.learnable_fc.weight"), t("param_kps_generator.learnable_fc.bias"),
                                    [-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], [0.08, 0.64])
    uv, _ = daf_prepare_ref.project_points(kp, t("projection_mat"), t("image_wh"))
    bs, cams, A, K, _ = uv.shape
    loc = uv.permute(0, 2, 3, 1, 4).reshape(bs, A * K, cams, 2).numpy()
    assert np.allclose(loc, d["call_sampling_location"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("xyz_act,scale_act", [("sigmoid", "sigmoid"), ("identity", "sigmoid"), ("sigmoid", "none"),
                                               ("identity", "identity")])
def test_key_point_restatement_activations_vs_the_reference_generator(xyz_act, scale_act):
    """The activation switches of the key-point generator (deformable_module.py:27-28, :66-67, :79-80): the
    reference's own class, executed here, against the restatement that checks gf_key_points on the GPU."""
    import torch
    import ref_shim
    from oracle import daf_prepare_ref
    if not ref_shim.available():
        pytest.skip("/root/reference absent (GPU box)")
    ref_shim.load_reference()
    fix = [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]]
    pc_range, scale_range = [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], [0.08, 0.64]
    gen = ref_shim.build_from_cfg(dict(type="SparseGaussian3DKeyPointsGenerator", embed_dims=16, num_learnable_pts=2,
                                       fix_scale=fix, pc_range=pc_range, scale_range=scale_range,
                                       xyz_activation=xyz_act, scale_activation=scale_act), ref_shim.MODELS).double()
    g = torch.Generator().manual_seed(11)
    anchor = torch.randn(2, 50, 12, generator=g, dtype=torch.float64)
    feat = torch.randn(2, 50, 16, generator=g, dtype=torch.float64)
    want = gen(anchor, feat)
    got = daf_prepare_ref.key_points(anchor, feat, fix, gen.learnable_fc.weight.detach(), gen.learnable_fc.bias.detach(),
                                     pc_range, scale_range, xyz_activation=xyz_act, scale_activation=scale_act)
    assert torch.allclose(got, want.detach(), rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------------------------------------ GPU
def _head_modules(name, d, gpu):
    import local_aggregate, local_aggregate_prob, local_aggregate_prob_fast
    prob = name != "base"
    pkg = {"base": local_aggregate, "prob": local_aggregate_prob, "prob_geosem": local_aggregate_prob,
           "prob_fast": local_aggregate_prob_fast}[name]
    return pkg.LocalAggregator(**_cuda_kwargs(d, prob)).to(gpu), prob


def _head_epilogue(name, outs, prob):
    """GaussianHead.forward :164-176 on the aggregator's outputs."""
    import torch
    if not prob:
        return outs[None].transpose(1, 2), None, None
    logits, bin_logits, density = outs
    if name == "prob_geosem":
        geosem = torch.cat([logits[:, :-1] * bin_logits.unsqueeze(-1), 1 - bin_logits.unsqueeze(-1)], dim=-1)
    else:
        geosem = logits
    return geosem[None].transpose(1, 2), bin_logits[None], density[None]


def _check(got, want, what, tol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-6)
    assert np.isfinite(got).all() and err <= tol, f"{what}: {err:.3e} > {tol}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", HEAD_CASES)
def test_head_call_replayed_on_the_dropin(gpu, name):
    """The exact module-level call the reference's GaussianHead made (aggregator inputs incl. the appended empty
    Gaussian and the LAPACK Sigma^-1), through the drop-in LocalAggregator on the HIP kernels."""
    import torch
    from gaussianformer_amd.head import occupancy_labels
    d = _load("head_" + name)
    agg, prob = _head_modules(name, d, gpu)
    t = lambda k, g=False: torch.from_numpy(d[k]).to(gpu).requires_grad_(g)
    means, opa, sem, cov = t("agg_in_means3D", True), t("agg_in_opacities", True), t("agg_in_semantics", True), t("agg_in_cov3D", True)
    outs = agg(t("agg_in_pts"), means, opa, sem, t("agg_in_scales"), cov)
    pred, bl, de = _head_epilogue(name, outs, prob)
    _check(pred.detach().cpu(), d["pred_occ"], "pred_occ", 1e-5)
    if prob:
        _check(bl.detach().cpu(), d["bin_logits"], "bin_logits", 1e-5)
        _check(de.detach().cpu(), d["density"], "density", 1e-5)
    # final_occ (gaussian_head.py:178-185) with the native epilogue kernel
    if prob:
        labels = occupancy_labels(outs[0], outs[1], threshold=0.5, empty_label=17, combine_geosem=name == "prob_geosem")
    else:
        labels = occupancy_labels(outs)
    assert np.array_equal(labels.cpu().numpy()[None], d["final_occ"])
    # raw-op arguments: the drop-in's host mirror hands the kernel what it handed the oracle when the fixture was made
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_forward
    call = [torch.from_numpy(d["call_" + k]).to(gpu) for k in ("pts", "points_int", "means3D", "means3D_int", "opacities",
                                                                "semantics", "radii", "cov3D")]
    raw = splat_forward(_lib.GF_SPLAT_PROB if prob else _lib.GF_SPLAT_BASE, *call, int(d["grid_H"]), int(d["grid_W"]), int(d["grid_D"]))
    _check(raw[0].cpu(), d["call_out_logits"], "raw logits", 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", HEAD_CASES)
def test_head_fused_path_matches_the_reference_caller(gpu, name):
    """From the head's own inputs (means, scales, rotations, opacities, semantics): ``GaussianArgs`` (the mirror of
    prepare_gaussian_args with the closed-form Sigma^-1) + the drop-in aggregator reproduce the reference caller's
    outputs and the gradients it sent back to every Gaussian property."""
    import torch
    from gaussianformer_amd.gaussian_prepare import GaussianArgs
    d = _load("head_" + name)
    agg, prob = _head_modules(name, d, gpu)
    args = GaussianArgs(num_classes=18, with_empty=not prob, use_localaggprob=prob,
                        empty_args=dict(mean=[0, 0, -1.0], scale=[100, 100, 8.0])).to(gpu)
    t = lambda k, g=False: torch.from_numpy(d[k]).to(gpu).requires_grad_(g)
    means, scales, rots, opa, sem_raw = t("means", True), t("scales", True), t("rotations", True), t("opacities", True), t("sem_raw", True)
    sem = torch.nn.functional.softplus(sem_raw)
    m, o, s, sc, cov = args(means, scales, rots, sem, opa)
    bs, g = m.shape[:2]
    outs = agg(t("occ_xyz").flatten(1, 3), m, o.reshape(bs, g), s, sc, cov)
    pred, bl, de = _head_epilogue(name, outs, prob)
    tol = 2e-3 if prob else 1e-4      # closed-form Sigma^-1 vs the reference's fp32 LAPACK inverse (prob: through the determinant)
    _check(pred.detach().cpu(), d["pred_occ"], "pred_occ", tol)
    loss = (pred * t("pred_weight")).sum()
    if prob:
        loss = loss + (bl * t("bin_weight")).sum() + (de * t("density_weight")).sum()
    loss.backward()
    gtol = 2e-2 if prob else 2e-3
    for k, leaf in (("means", means), ("scales", scales), ("rotations", rots), ("opacities", opa), ("sem_raw", sem_raw)):
        _check(leaf.grad.cpu(), d["grad_" + k], "grad_" + k, gtol)
    if not prob:
        _check(args.empty_scalar.grad.cpu(), d["grad_empty_scalar"], "grad_empty_scalar", gtol)


@pytest.mark.gpu
def test_dfa_call_replayed_on_the_dropin(gpu):
    """DeformableFeatureAggregation.forward: the raw op with the arguments the reference caller produced, and the
    whole module body rebuilt from the product's pieces (feature_maps_format, deformable_prepare, DAF.apply) with the
    reference's parameters -- output and gradients against what the reference computed."""
    import torch
    import torch.nn.functional as F
    from oracle import daf_prepare_ref
    from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
    from gaussianformer_amd.deformable_prepare import deformable_prepare
    d = _load("dfa")
    t = lambda k, g=False: torch.from_numpy(d[k]).to(gpu).requires_grad_(g)
    # (1) raw op
    out = DAF.apply(t("call_mc_ms_feat"), t("call_spatial_shape"), t("call_scale_start_index"), t("call_sampling_location"),
                    t("call_weights"))
    _check(out.cpu(), d["call_out_output"], "raw daf output", 1e-5)
    # (2) module body
    C, G, cams = int(d["embed_dims"]), int(d["num_groups"]), int(d["num_cams"])
    L = len(d["levels"])
    P = {k[len("param_"):]: t(k, True) for k in d.files if k.startswith("param_")}
    inst, emb, anchor = t("instance_feature", True), t("anchor_embed", True), t("anchor", True)
    fmaps = [t(f"feature_map{i}", True) for i in range(L)]
    pm, wh = t("projection_mat"), t("image_wh")
    bs, A = inst.shape[:2]
    kp = daf_prepare_ref.key_points(anchor, inst, [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0],
                                                   [0, 0, 0.45], [0, 0, -0.45]],
                                    P["kps_generator.learnable_fc.weight"], P["kps_generator.learnable_fc.bias"],
                                    [-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], [0.08, 0.64])
    K = kp.shape[2]
    table = DAF.feature_maps_format(fmaps)
    # _get_weights (:250-279): camera embedding (linear_relu_ln(embed_dims, 1, 2, 12)) + weights_fc
    cam = pm[:, :, :3].reshape(bs, cams, -1)
    x = cam
    layers = sorted({int(k.split(".")[1]) for k in P if k.startswith("camera_encoder.")})
    for i in layers:
        w = P[f"camera_encoder.{i}.weight"]
        if w.dim() == 2:
            x = F.relu(F.linear(x, w, P[f"camera_encoder.{i}.bias"]))
        else:
            x = F.layer_norm(x, (w.shape[0],), w, P[f"camera_encoder.{i}.bias"])
    feature = (inst + emb)[:, :, None] + x[:, None]
    raw = F.linear(feature, P["weights_fc.weight"], P["weights_fc.bias"]).reshape(bs, A, cams, L, K, G)
    loc, weights = deformable_prepare(kp, pm, wh, raw)
    feats = DAF.apply(*table, loc, weights).reshape(bs, A, K, C).sum(dim=2)
    output = torch.cat([F.linear(feats, P["output_proj.weight"], P["output_proj.bias"]), inst], dim=-1)
    _check(output.detach().cpu(), d["output"], "module output", 1e-4)
    (output * t("out_weight")).sum().backward()
    for k, leaf in (("instance_feature", inst), ("anchor_embed", emb), ("anchor", anchor)):
        _check(leaf.grad.cpu(), d["grad_" + k], "grad_" + k, 2e-3)
    for i, f in enumerate(fmaps):
        _check(f.grad.cpu(), d[f"grad_feature_map{i}"], f"grad_feature_map{i}", 2e-3)
    for k, p in P.items():
        if "grad_param_" + k in d.files:
            _check(p.grad.cpu(), d["grad_param_" + k], "grad_param_" + k, 2e-3)
