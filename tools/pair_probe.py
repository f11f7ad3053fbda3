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


def run(si, flags, env=None, steps=0):
    for k in ("GF_MFMA_WAVE", "GF_MFMA_TILE"):
        os.environ.pop(k, None)
    if env:
        os.environ[env] = "1"
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
    o_pair, w_pair, t_pair, r_pair = run(si, 0, None, steps)
    o_wave, w_wave, t_wave, r_wave = run(si, 0, "GF_MFMA_WAVE", steps)
    o_ex, w_ex, t_ex, _ = run(si, _lib.GF_EXACT_FP32, None, 0)
    print(f"{config} P={si.means3D.shape[0]} grid {si.H}x{si.W}x{si.D}: paths pair {w_pair} wave {w_wave} exact {w_ex}; "
          f"reproducible pair {r_pair} wave {r_wave}; us/step pair {t_pair} wave {t_wave}", flush=True)
    print(f"   pair vs wave (scaled, abs) {err(o_pair, o_wave)}; pair vs exact {err(o_pair, o_ex)}; wave vs exact {err(o_wave, o_ex)}; "
          f"finite {bool(torch.isfinite(o_pair).all())}", flush=True)
    if not kw:
        try:
            from oracle import ref
            pi, mi, radii, cov6 = prep(si)
            r = torch.from_numpy(ref.splat_forward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"])
            print(f"   vs oracle/_ref (scaled, abs): pair {err(o_pair.cpu(), r)} wave {err(o_wave.cpu(), r)} exact {err(o_ex.cpu(), r)}", flush=True)
        except Exception as exc:
            print("   no oracle/_ref:", exc)
