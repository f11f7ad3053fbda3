"""Development probe (GPU box): error of the HIP prob splat against oracle/_ref for every exp flavour and
both determinant choices.  python tools/ref_probe.py [full]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import ref
from gaussianformer_amd import _lib
from gaussianformer_amd.synthetic import make_splat_inputs
from util import hip_splat_forward, hip_splat_backward, prep

dev = torch.device("cuda:0")
full = len(sys.argv) > 1 and sys.argv[1] == "full"
cases = [("prob_gs6400", dict(seed=0), False), ("prob_gs6400", dict(seed=0), True)] if full else \
        [("prob_gs6400", dict(seed=47, P=120, H=24, W=20, D=16), False), ("prob_gs6400", dict(seed=3, P=400, H=40, W=40, D=16), True)]
for config, kw, per_axis in cases:
    si = make_splat_inputs(config, **kw)
    pi, mi, radii, cov6 = prep(si, per_axis)
    rng = np.random.default_rng(5)
    N = si.pts.shape[0]
    g, gb, gd = rng.standard_normal((N, 18)).astype(np.float32), rng.standard_normal(N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    rf, rg, _ = ref.splat_forward_backward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D, g, gb, gd)
    fin = np.isfinite(rf["logits"]).all(axis=1) & np.isfinite(rf["probability"])
    print(f"== {config} {kw} per_axis={per_axis}: R={rf['num_rendered']} non-finite ref voxels {(~fin).sum()}")
    for ename, ef in (("fast", 0), ("comp", _lib.GF_COMP_EXP), ("libm", _lib.GF_LIBM_EXP)):
        for dname, df in (("det32", 0), ("det64", _lib.GF_PROB_EXACT_DET)):
            got, t, state, fwd_t = hip_splat_forward(dev, si, pi, mi, radii, cov6, flags=ef | df)
            errs = []
            for k in ("logits", "bin_logits", "density", "probability"):
                a, b = got[k][fin], rf[k][fin]
                errs.append(float((np.abs(a.astype(np.float64) - b) / np.maximum(1.0, np.abs(b))).max()))
            grads = hip_splat_backward(dev, si, t, state, fwd_t, g, gb, gd, flags=ef | df)
            gerrs = []
            for a, b in zip(grads, rg):
                ok = np.isfinite(b) if b.ndim == 1 else np.isfinite(b).all(axis=1)
                gerrs.append(float(np.abs(a[ok].astype(np.float64) - b[ok]).max() / max(np.abs(b[ok]).max(), 1e-6)))
            print(f"  exp={ename} {dname}: fwd " + " ".join(f"{e:.2e}" for e in errs) + " | bwd " + " ".join(f"{e:.2e}" for e in gerrs))
