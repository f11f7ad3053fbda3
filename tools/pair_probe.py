"""Development probe (GPU box): the pair kernel (default) against the wave kernel (GF_MFMA_WAVE=1), the exact-fp32 kernel and
oracle/_ref -- errors (scaled and absolute), run-to-run reproducibility, step times.  python tools/pair_probe.py [config ...]"""
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
args = [a for a in sys.argv[1:] if not a.startswith("--")]
small = "--small" in sys.argv
configs = args or ["nuscenes_gs25600_solid"]


_OPTS = {"GF_MFMA_SOLO": "dev.splat_solo", "GF_MFMA_TILE": "splat.mfma_tile_kernel", "GF_MFMA_PAIR": "dev.splat_pair",
         "GF_SOLO_WAVES": "dev.splat_solo_waves", "GF_FUSED": "dev.splat_fused"}


def run(si, flags, env=None, steps=0):
    for k in _OPTS.values():   # (development build: GF_LIB=.../libgf_hip_dev.so)
        _lib.set_option(k, 0)
    if env:
        for e in env.split(","):
            k, _, v = e.partition("=")
            _lib.set_option(_OPTS[k], int(v or "1"))
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=flags)
    out = plan.run().clone()
    torch.cuda.synchronize()
    words = plan.state_words()[:3]
    us = None
    if steps:
        for _ in range(30):
            plan.run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(steps):
                plan.run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e6)
        us = sorted(ts)[2]
    out2 = plan.run().clone()
    torch.cuda.synchronize()
    return out, words, us, bool(torch.equal(out, out2))


def err(a, b):
    a, b = a.double(), b.double()
    d = (a - b).abs()
    return float((d / b.abs().clamp(min=1.0)).max()), float(d.max())


cases = []
for config in configs:
    cases.append((config, dict()))
if small:
    for seed, (P, H, W, D) in enumerate([(300, 16, 16, 8), (2000, 40, 40, 16), (777, 20, 36, 12), (5000, 64, 48, 8), (64, 8, 8, 4), (1, 8, 8, 8), (3000, 30, 50, 16)]):
        cases.append(("nuscenes_gs25600_solid", dict(P=P, H=H, W=W, D=D, seed=seed + 1)))
for config, kw in cases:
    si = make_splat_inputs(config, seed=kw.pop("seed", 0), **kw)
    steps = 200 if not kw else 0
    res = {}
    for name, flags, env in (("fused", 1, "GF_FUSED"), ("fused3", 1, "GF_FUSED,GF_SOLO_WAVES=3"), ("solo", 0, "GF_MFMA_SOLO"), ("solo3", 0, "GF_MFMA_SOLO,GF_SOLO_WAVES=3"), ("wave", 0, None), ("wave_dense", 1, None), ("exact", _lib.GF_EXACT_FP32, None)):
        res[name] = run(si, flags, env, steps if name != "exact" else 0)
    print(f"{config} P={si.means3D.shape[0]} grid {si.H}x{si.W}x{si.D}: " + "; ".join(f"{k} path {v[1][1]} bits {v[1][2]:#x} repro {v[3]}" + (f" {v[2]:.2f} us" if v[2] else "") for k, v in res.items()), flush=True)
    ex = res["exact"][0]
    print("   vs exact (scaled, abs): " + "; ".join(f"{k} ({err(v[0], ex)[0]:.2e}, {err(v[0], ex)[1]:.2e})" for k, v in res.items() if k != "exact") +
          f"; solo3 == solo {bool(torch.equal(res['solo'][0], res['solo3'][0]))}; finite {bool(torch.isfinite(res['solo'][0]).all())}", flush=True)
    if not kw:
        try:
            from oracle import ref
            pi, mi, radii, cov6 = prep(si)
            r = torch.from_numpy(ref.splat_forward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"])
            print("   vs oracle/_ref (scaled, abs): " + "; ".join(f"{k} ({err(v[0].cpu(), r)[0]:.2e}, {err(v[0].cpu(), r)[1]:.2e})" for k, v in res.items()), flush=True)
        except Exception as exc:
            print("   no oracle/_ref:", exc)
