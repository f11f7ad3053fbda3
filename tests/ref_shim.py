"""TEST INFRASTRUCTURE: run the reference's own CALLER code (``GaussianHead``,
``DeformableFeatureAggregation``, ``SparseGaussian3DKeyPointsGenerator``) against this repository's drop-in
packages, in a process where mmengine / mmseg are not installed.

* ``install_stubs()`` puts minimal stand-ins for the three third-party names those files import into
  ``sys.modules``: ``mmengine.registry.MODELS`` / ``mmseg.models.HEADS`` (a registry with ``register_module`` and
  ``build``), ``mmengine.build_from_cfg``, ``mmengine.model.BaseModule`` (+ ``xavier_init`` / ``constant_init``).
* ``load_reference(root)`` imports ``model/head/gaussian_head.py`` and
  ``model/encoder/gaussian_encoder/deformable_module.py`` **from the reference tree, unmodified**, under a
  synthetic package ``gf_refmodel`` whose ``__init__`` files are not executed (they would pull in mmseg
  backbones); the relative import ``from .ops import DeformableAggregationFunction as DAF``
  (deformable_module.py:11-14) is pointed at this repository's drop-in ``model.encoder.gaussian_encoder.ops``,
  and ``import local_aggregate[_prob[_fast]]`` (gaussian_head.py:30-39) resolves to the drop-in packages at the
  repo root, exactly as it would in a deployment.
* ``cpu_kernels()`` swaps the raw ops of the product (which have no CPU path) for oracle-backed CPU functions,
  so the reference callers can be executed here, without a GPU, to record fixtures
  (tools/make_golden_callers.py).  Only the four kernel entry points are replaced: the host mirrors'
  own code (integer path, covariance packing, autograd routing, dtype coercions) is what runs.

Needs ``/root/reference``; never imported by the product.
"""
import contextlib
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE = os.environ.get("GF_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE, "model", "head", "gaussian_head.py"))


class _Registry:
    def __init__(self, name):
        self.name = name
        self.modules = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.modules[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self.modules[key]

    def build(self, cfg, **kw):
        return build_from_cfg(cfg, self, **kw)


def build_from_cfg(cfg, registry, default_args=None):
    cfg = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            cfg.setdefault(k, v)
    cls = registry.get(cfg.pop("type"))
    return cls(**cfg)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


def _xavier_init(module, gain=1, bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


MODELS = _Registry("model")
HEADS = MODELS   # mmseg's HEADS is a child of the same root registry


def install_stubs():
    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None or not getattr(m, "_gf_stub", False):
            m = types.ModuleType(name)
            m._gf_stub = True
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m
    for name in ("mmengine", "mmseg"):
        if name in sys.modules and not getattr(sys.modules[name], "_gf_stub", False):
            raise RuntimeError(f"a real {name} is installed: the stubs are not needed")
    reg = mod("mmengine.registry", MODELS=MODELS)
    model = mod("mmengine.model", BaseModule=BaseModule, xavier_init=_xavier_init, constant_init=_constant_init)
    mod("mmengine", build_from_cfg=build_from_cfg, registry=reg, model=model)
    mm = mod("mmseg.models", HEADS=HEADS)
    mod("mmseg", models=mm)


def load_reference(root=None):
    """Returns a namespace with the reference's ``GaussianHead``, ``DeformableFeatureAggregation``,
    ``SparseGaussian3DKeyPointsGenerator`` and ``GaussianPrediction`` classes, wired to the drop-ins."""
    root = root or REFERENCE
    install_stubs()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]           # a package whose __init__.py is NOT executed
        m.__package__ = name
        sys.modules[name] = m
        return m
    base = "gf_refmodel"
    pkg(base, os.path.join(root, "model"))
    pkg(base + ".head", os.path.join(root, "model", "head"))
    pkg(base + ".utils", os.path.join(root, "model", "utils"))
    pkg(base + ".encoder", os.path.join(root, "model", "encoder"))
    pkg(base + ".encoder.gaussian_encoder", os.path.join(root, "model", "encoder", "gaussian_encoder"))
    # the drop-in for `from .ops import DeformableAggregationFunction as DAF`
    sys.modules[base + ".encoder.gaussian_encoder.ops"] = importlib.import_module("model.encoder.gaussian_encoder.ops")
    head = importlib.import_module(base + ".head.gaussian_head")
    dfa = importlib.import_module(base + ".encoder.gaussian_encoder.deformable_module")
    enc_utils = importlib.import_module(base + ".encoder.gaussian_encoder.utils")
    assert dfa.DAF is not None, "the drop-in ops package did not import"
    return types.SimpleNamespace(GaussianHead=head.GaussianHead, DeformableFeatureAggregation=dfa.DeformableFeatureAggregation,
                                 SparseGaussian3DKeyPointsGenerator=dfa.SparseGaussian3DKeyPointsGenerator,
                                 GaussianPrediction=enc_utils.GaussianPrediction, DAF=dfa.DAF, head_module=head, dfa_module=dfa)


# ---------------------------------------------------------------------------------------------------------------
# oracle-backed CPU stand-ins for the product's four raw ops
# ---------------------------------------------------------------------------------------------------------------

def _np(t):
    return None if t is None else t.detach().cpu().numpy()


@contextlib.contextmanager
def cpu_kernels(record=None):
    """Within the context the raw ops of ``gaussianformer_amd`` run on the CPU through the oracle.  ``record`` (a
    list) receives one dict per raw-op call with the arguments the host mirror handed to the kernel."""
    import oracle
    from gaussianformer_amd import _lib, deformable_aggregation as da, local_aggregate as la

    def splat_forward(variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D, H, W, D, flags=0):
        v = "prob" if variant == _lib.GF_SPLAT_PROB else "base"
        if record is not None:
            record.append(dict(op="splat_forward", variant=v, pts=_np(pts), points_int=_np(points_int), means3D=_np(means3D),
                               means3D_int=_np(means3D_int), opacities=_np(opacities), semantics=_np(semantics),
                               radii=_np(radii), cov3D=_np(cov3D), H=H, W=W, D=D))
        r = oracle.splat_forward(v, _np(pts), _np(points_int), _np(means3D), _np(means3D_int), _np(opacities),
                                 _np(semantics), _np(radii), _np(cov3D), H, W, D)
        t = lambda a: None if a is None else torch.from_numpy(a)
        if record is not None:
            record[-1].update({"out_" + k: v for k, v in r.items() if isinstance(v, np.ndarray)})
        out = (t(r["logits"]), t(r.get("bin_logits")), t(r.get("density")), t(r.get("probability")), torch.zeros(1))
        if flags & _lib.GF_PROB_NUMERATOR:
            raise NotImplementedError
        return out

    def splat_backward(variant, pts, points_int, means3D, means3D_int, opacities, semantics, radii, cov3D, H, W, D,
                       logits_grad, fwd_outputs=None, bin_logits_grad=None, density_grad=None, state=None, flags=0):
        v = "prob" if variant == _lib.GF_SPLAT_PROB else "base"
        fwd = None
        if v == "prob":
            fwd = dict(zip(("logits", "bin_logits", "density", "probability"), (_np(x) for x in fwd_outputs)))
            N = pts.shape[0]
            bin_logits_grad = np.zeros(N, np.float32) if bin_logits_grad is None else _np(bin_logits_grad)
            density_grad = np.zeros(N, np.float32) if density_grad is None else _np(density_grad)
        g = oracle.splat_backward(v, _np(pts), _np(points_int), _np(means3D), _np(means3D_int), _np(opacities),
                                  _np(semantics), _np(radii), _np(cov3D), H, W, D, _np(logits_grad), fwd=fwd,
                                  bin_grad=bin_logits_grad, density_grad=density_grad)
        return tuple(torch.from_numpy(x) for x in g)

    def daf_forward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights):
        if record is not None:
            record.append(dict(op="daf_forward", mc_ms_feat=_np(mc_ms_feat), spatial_shape=_np(spatial_shape),
                               scale_start_index=_np(scale_start_index), sampling_location=_np(sampling_location),
                               weights=_np(weights)))
        out = oracle.daf_forward(_np(mc_ms_feat), _np(spatial_shape), _np(scale_start_index), _np(sampling_location),
                                 _np(weights))
        if record is not None:
            record[-1]["out_output"] = out
        return torch.from_numpy(out)

    def daf_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, grad_output,
                     grad_mc_ms_feat, grad_sampling_location, grad_weights, pixel_major=True):
        gf, gl, gw = oracle.daf_backward(_np(mc_ms_feat), _np(spatial_shape), _np(scale_start_index),
                                         _np(sampling_location), _np(weights), _np(grad_output))
        grad_mc_ms_feat += torch.from_numpy(gf)
        grad_sampling_location += torch.from_numpy(gl)
        grad_weights += torch.from_numpy(gw)

    saved = (la.splat_forward, la.splat_backward, da.deformable_aggregation_forward, da.deformable_aggregation_backward)
    la.splat_forward, la.splat_backward = splat_forward, splat_backward
    da.deformable_aggregation_forward, da.deformable_aggregation_backward = daf_forward, daf_backward
    try:
        yield
    finally:
        la.splat_forward, la.splat_backward, da.deformable_aggregation_forward, da.deformable_aggregation_backward = saved
