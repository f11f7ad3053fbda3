"""Development probe (GPU box): the matrix-core render kernel (the default) against the exact-fp32 tile kernel and oracle/_ref.
python tools/mfma_probe.py [config ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev

dev = torch.device("cuda:0")
configs = sys.argv[1:] or ["nuscenes_gs25600_solid", "nuscenes_gs144000"]
for config in configs:
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    outs = {}
    for name, flags in (("exact", _lib.GF_EXACT_FP32), ("mfma", 0)):
        plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=flags)
        outs[name] = plan.run().clone()
        torch.cuda.synchronize()
        print(f"{config} {name}: state words {plan.state_words()}", flush=True)
        for _ in range(20):
            plan.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            plan.run()
        torch.cuda.synchronize()
        print(f"{config} {name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step", flush=True)
    a, b = outs["exact"].double(), outs["mfma"].double()
    err = ((a - b).abs() / a.abs().clamp(min=1.0)).max().item()
    print(f"{config}: mfma vs exact max scaled err {err:.3e}, max ABSOLUTE err {(a - b).abs().max().item():.3e} (max |logit| {a.abs().max().item():.2f}); "
          f"finite {bool(torch.isfinite(outs['mfma']).all())}")
    try:
        from oracle import ref
        if ref.available() or os.path.isdir(os.path.join(ROOT, "oracle", "_ref")):
            r = ref.splat_forward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"].astype(np.float64)
            for name in outs:
                g = outs[name].cpu().numpy().astype(np.float64)
                print(f"{config}: {name} vs oracle/_ref max scaled err {(np.abs(g - r) / np.maximum(1.0, np.abs(r))).max():.3e}, "
                      f"max ABSOLUTE err {np.abs(g - r).max():.3e} (max |logit| {np.abs(r).max():.2f})")
    except Exception as exc:
        print("no oracle/_ref:", exc)
