"""Generates tests/golden/caller_*.npz by executing the REFERENCE's own caller code, unmodified, against this
repository's drop-in packages (build container only; needs /root/reference, no GPU):

* ``GaussianHead.forward`` (model/head/gaussian_head.py:122-197, incl. ``prepare_gaussian_args`` :82-120) in four
  flavours -- base with the appended empty Gaussian, prob (threshold epilogue), prob + ``combine_geosem``,
  prob_fast -- constructed from config-shaped kwargs (``cuda_kwargs``, ``empty_args`` … of
  config/nuscenes_gs25600_solid.py:174-191 and config/prob/nuscenes_gs6400.py) on a reduced grid;
* ``DeformableFeatureAggregation.forward`` (model/encoder/gaussian_encoder/deformable_module.py:146-248) with its
  ``SparseGaussian3DKeyPointsGenerator``, built through the registry from a config-shaped dict.

mmengine / mmseg are replaced by the stand-ins of tests/ref_shim.py; ``import local_aggregate*`` and
``from .ops import DeformableAggregationFunction`` resolve to the drop-ins, whose host-side code (integer path,
covariance packing, dtype coercions, autograd routing, ``feature_maps_format``) runs as shipped; only the four raw
kernel entry points are served by the CPU oracle (``ref_shim.cpu_kernels``), since the product has no CPU path and
this container has no GPU.  ``Tensor.cuda()`` (gaussian_head.py:119) is an identity in this process.

Each fixture holds: the module inputs, the arguments the reference caller handed to the raw op (as the drop-in's
host mirror coerced them), the module outputs and the gradients of a fixed scalar loss with respect to every
differentiable input.  tests/test_ref_callers.py replays them on the GPU.

Run:  python tools/make_golden_callers.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
GRID = dict(H=16, W=16, D=8, pc_min=[-4.0, -4.0, -2.0], grid_size=0.5)


def voxel_centres():
    H, W, D, gs, lo = GRID["H"], GRID["W"], GRID["D"], GRID["grid_size"], GRID["pc_min"]
    ax = [(np.arange(n, dtype=np.float32) + np.float32(0.5)) * np.float32(gs) + np.float32(o) for n, o in zip((H, W, D), lo)]
    return np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1)[None].astype(np.float32)      # [1,H,W,D,3]


def head_case(ref, name, g, seed, scale_range, head_kwargs):
    rng = np.random.default_rng(seed)
    lo = np.array(GRID["pc_min"])
    ext = np.array([GRID["H"], GRID["W"], GRID["D"]]) * GRID["grid_size"]
    leaf = lambda a: torch.tensor(a, dtype=torch.float32, requires_grad=True)
    means = leaf((lo + (0.02 + 0.96 * rng.random((1, g, 3))) * ext))
    scales = leaf(scale_range[0] + (scale_range[1] - scale_range[0]) * rng.random((1, g, 3)))
    rotations = leaf(rng.standard_normal((1, g, 4)))   # un-normalised: get_rotation_matrix is applied to it as is
    with torch.no_grad():
        rotations /= rotations.norm(dim=-1, keepdim=True)
    opacities = leaf(rng.random((1, g, 1)))
    sem_raw = leaf(rng.standard_normal((1, g, 17)))
    semantics = torch.nn.functional.softplus(sem_raw)
    head = ref.GaussianHead(**head_kwargs)
    head.train()
    metas = dict(occ_xyz=torch.from_numpy(voxel_centres()),
                 occ_label=torch.from_numpy(rng.integers(0, 18, (1, GRID["H"], GRID["W"], GRID["D"]))),
                 occ_cam_mask=torch.ones(1, GRID["H"], GRID["W"], GRID["D"], dtype=torch.bool))
    rep = [dict(gaussian=ref.GaussianPrediction(means=means, scales=scales, rotations=rotations, opacities=opacities,
                                                semantics=semantics))]
    calls, agg_in = [], []
    head.aggregator.register_forward_pre_hook(lambda m, a: agg_in.append([t.detach().numpy().copy() for t in a]))
    with ref_shim.cpu_kernels(record=calls):
        out = head(rep, metas)
        pred = out["pred_occ"][-1]                                  # [1, 18, N]
        w = torch.from_numpy(rng.standard_normal(tuple(pred.shape)).astype(np.float32))
        loss = (pred * w).sum()
        wb = wd = None
        if out["bin_logits"]:
            wb = torch.from_numpy(rng.standard_normal(tuple(out["bin_logits"][-1].shape)).astype(np.float32))
            wd = torch.from_numpy(rng.standard_normal(tuple(out["density"][-1].shape)).astype(np.float32))
            loss = loss + (out["bin_logits"][-1] * wb).sum() + (out["density"][-1] * wd).sum()
        loss.backward()
    assert len(calls) == 1 and calls[0]["op"] == "splat_forward"
    call = calls[0]
    d = dict(kind="GaussianHead.forward", variant=call["variant"], **{f"grid_{k}": v for k, v in GRID.items()},
             means=means.detach().numpy(), scales=scales.detach().numpy(), rotations=rotations.detach().numpy(),
             opacities=opacities.detach().numpy(), semantics=semantics.detach().numpy(), sem_raw=sem_raw.detach().numpy(),
             occ_xyz=metas["occ_xyz"].numpy(), pred_weight=w.numpy(),
             pred_occ=pred.detach().numpy(), final_occ=out["final_occ"].numpy(),
             grad_means=means.grad.numpy(), grad_scales=scales.grad.numpy(), grad_rotations=rotations.grad.numpy(),
             grad_opacities=opacities.grad.numpy(), grad_sem_raw=sem_raw.grad.numpy(),
             **{"call_" + k: v for k, v in call.items() if isinstance(v, np.ndarray)},
             **{"agg_in_" + k: v for k, v in zip(("pts", "means3D", "opacities", "semantics", "scales", "cov3D"), agg_in[0])})
    if wb is not None:
        d.update(bin_weight=wb.numpy(), density_weight=wd.numpy(), bin_logits=out["bin_logits"][-1].detach().numpy(),
                 density=out["density"][-1].detach().numpy())
    if head_kwargs.get("with_empty"):
        d.update(grad_empty_scalar=head.empty_scalar.grad.numpy())
    assert all(np.isfinite(v).all() for v in d.values() if isinstance(v, np.ndarray) and v.dtype.kind == "f"), name
    np.savez_compressed(os.path.join(OUT, f"caller_head_{name}.npz"), **d)
    print(name, "pred_occ", tuple(pred.shape), "state_dict keys", sorted(head.state_dict().keys()))


def dfa_case(ref):
    rng = np.random.default_rng(7)
    torch.manual_seed(7)
    bs, A, C, G, cams = 1, 40, 32, 4, 3
    levels = ((12, 20), (6, 10), (3, 5))
    cfg = dict(type="DeformableFeatureAggregation", embed_dims=C, num_groups=G, num_levels=len(levels), num_cams=cams,
               attn_drop=0.0, use_deformable_func=True, use_camera_embed=True, residual_mode="cat",
               kps_generator=dict(type="SparseGaussian3DKeyPointsGenerator", embed_dims=C, num_learnable_pts=2,
                                  fix_scale=[[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0],
                                             [0, 0, 0.45], [0, 0, -0.45]],
                                  pc_range=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], scale_range=[0.08, 0.64]))
    dfa = ref_shim.build_from_cfg(cfg, ref_shim.MODELS)
    dfa.eval()
    leaf = lambda a: torch.tensor(a, dtype=torch.float32, requires_grad=True)
    instance_feature = leaf(rng.standard_normal((bs, A, C)))
    anchor_embed = leaf(rng.standard_normal((bs, A, C)))
    anchor = leaf(rng.standard_normal((bs, A, 28)))        # xyz(3) scale(3) rot(4) opa(1) sem(17), pre-activation
    fmaps = [leaf(rng.standard_normal((bs, cams, C, h, w))) for h, w in levels]
    mats = []
    for c in range(cams):
        yaw = 2 * np.pi * c / cams
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0]); up = np.array([0.0, 0.0, 1.0]); right = np.cross(fwd, up)
        R = np.stack([right, -up, fwd])
        t = -R @ np.array([0.0, 0.0, 1.5])
        K = np.array([[300.0, 0, 400.0], [0, 300.0, 225.0], [0, 0, 1.0]])
        M = np.eye(4); M[:3, :4] = K @ np.concatenate([R, t[:, None]], axis=1)
        mats.append(M)
    metas = dict(projection_mat=torch.tensor(np.stack(mats)[None], dtype=torch.float32),
                 image_wh=torch.tensor([[[800.0, 450.0]] * cams], dtype=torch.float32))
    calls = []
    with ref_shim.cpu_kernels(record=calls):
        out = dfa(instance_feature, anchor, anchor_embed, fmaps, metas)
        w = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32))
        (out * w).sum().backward()
    assert len(calls) == 1 and calls[0]["op"] == "daf_forward"
    call = calls[0]
    seen = float((call["weights"].sum(axis=(2, 3)) > 0).mean())
    d = dict(kind="DeformableFeatureAggregation.forward", levels=np.array(levels), embed_dims=C, num_groups=G, num_cams=cams,
             instance_feature=instance_feature.detach().numpy(), anchor_embed=anchor_embed.detach().numpy(),
             anchor=anchor.detach().numpy(), projection_mat=metas["projection_mat"].numpy(), image_wh=metas["image_wh"].numpy(),
             out_weight=w.numpy(), output=out.detach().numpy(),
             grad_instance_feature=instance_feature.grad.numpy(), grad_anchor_embed=anchor_embed.grad.numpy(),
             grad_anchor=anchor.grad.numpy(),
             **{f"feature_map{i}": f.detach().numpy() for i, f in enumerate(fmaps)},
             **{f"grad_feature_map{i}": f.grad.numpy() for i, f in enumerate(fmaps)},
             **{"param_" + k: v.detach().numpy() for k, v in dfa.state_dict().items()},
             **{"grad_param_" + k: p.grad.numpy() for k, p in dfa.named_parameters() if p.grad is not None},
             **{"call_" + k: v for k, v in call.items() if isinstance(v, np.ndarray)})
    np.savez_compressed(os.path.join(OUT, "caller_dfa.npz"), **d)
    print("dfa output", tuple(out.shape), "fraction of sample points seen by a camera", seen)


if __name__ == "__main__":
    assert ref_shim.available(), "needs /root/reference"
    torch.Tensor.cuda = lambda self, *a, **k: self      # gaussian_head.py:119 `.cpu().inverse().cuda()` on a CPU-only box
    ref = ref_shim.load_reference()
    cuda_kwargs = dict(scale_multiplier=3, **GRID)
    head_case(ref, "base", g=60, seed=11, scale_range=(0.08, 0.64), head_kwargs=dict(
        apply_loss_type="random_1", num_classes=18, empty_args=dict(mean=[0, 0, -1.0], scale=[100, 100, 8.0]),
        with_empty=True, cuda_kwargs=cuda_kwargs))
    prob_kwargs = dict(scale_multiplier=4, **GRID)
    head_case(ref, "prob", g=40, seed=12, scale_range=(0.05, 1.2), head_kwargs=dict(
        apply_loss_type="random_1", num_classes=18, with_empty=False, use_localaggprob=True, cuda_kwargs=prob_kwargs))
    head_case(ref, "prob_geosem", g=40, seed=13, scale_range=(0.05, 1.2), head_kwargs=dict(
        apply_loss_type="random_1", num_classes=18, with_empty=False, use_localaggprob=True, combine_geosem=True,
        cuda_kwargs=prob_kwargs))
    head_case(ref, "prob_fast", g=40, seed=14, scale_range=(0.05, 1.2), head_kwargs=dict(
        apply_loss_type="random_1", num_classes=18, with_empty=False, use_localaggprob=True, use_localaggprob_fast=True,
        cuda_kwargs=prob_kwargs))
    dfa_case(ref)
