"""Development probe (GPU box): the matrix-core backward (default after a matrix-core forward) against the exact Gaussian-major
kernels (GF_EXACT_FP32) and, when built, the reference's own kernels.   python tools/bwd_probe.py [small|full] [config ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
from util import grad_row_errors, prep, to_dev, whole_grid_rows

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "small"
configs = sys.argv[2:] or ["nuscenes_gs25600_solid", "nuscenes_gs144000"]
names = ("means", "opacity", "semantics", "cov")


def run(si, tag, grad_scale=1.0):
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    logits, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
    torch.cuda.synchronize()
    words = state.view(torch.int32)[:3].tolist()
    g = (torch.randn(logits.shape, generator=torch.Generator().manual_seed(1)) * grad_scale).to(dev)
    outs = {}
    for name, flags in (("exact", _lib.GF_EXACT_FP32), ("auto", 0), ("mfma", _lib.GF_MFMA_SPLAT)):
        outs[name] = [x.clone() for x in splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=flags)]
        torch.cuda.synchronize()
    print(f"{tag}: forward state words {words}", flush=True)
    for name in ("auto", "mfma"):
        for k, (a, b) in enumerate(zip(outs["exact"], outs[name])):
            a64, b64 = a.double(), b.double()
            scale = float(a64.abs().max()) + 1e-30
            err = float((a64 - b64).abs().max()) / scale
            fin = bool(torch.isfinite(b).all())
            line = f"{tag}: {name:5s} {names[k]:10s} max err / max|exact| = {err:.3e}  (absolute {err * scale:.3e}, max|exact| {scale:.3e})  finite {fin}"
            if err > 1e-3 or not fin:
                d = (a64 - b64).abs().reshape(a.shape[0], -1)
                worst = int(d.max(dim=1)[0].argmax())
                colerr = (d.max(dim=0)[0] / scale).tolist()
                line += f"  worst Gaussian {worst}: exact {a.reshape(a.shape[0], -1)[worst].tolist()[:6]} got {b.reshape(b.shape[0], -1)[worst].tolist()[:6]}; per-column err {[f'{c:.1e}' for c in colerr]}"
            print(line, flush=True)
    try:
        from oracle import ref
        if ref.available():
            rg = ref.splat_backward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D, g.cpu().numpy())
            for name in ("exact", "auto"):
                errs = [float(np.abs(o.cpu().numpy().reshape(w.shape).astype(np.float64) - w).max() / max(np.abs(w).max(), 1e-30)) for o, w in zip(outs[name], rg)]
                aerr = [float(np.abs(o.cpu().numpy().reshape(w.shape).astype(np.float64) - w).max()) for o, w in zip(outs[name], rg)]
                print(f"{tag}: {name:5s} vs oracle/_ref (scaled by max|ref| / absolute): " + ", ".join(f"{n} {e:.2e} / {ae:.2e}" for n, e, ae in zip(names, errs, aerr)), flush=True)
                whole = whole_grid_rows(mi, radii, si.H, si.W, si.D)
                rows = [grad_row_errors(o.cpu().numpy().reshape(w.shape), w, whole) for o, w in zip(outs[name], rg)]
                print(f"{tag}: {name:5s} vs oracle/_ref ROW BY ROW (worst ordinary row / whole-grid row; tests/util.py): "
                      + ", ".join(f"{n} {e['ordinary']:.2e} / {e['whole_grid']:.2e}" for n, e in zip(names, rows)), flush=True)
    except Exception as exc:
        print("no oracle/_ref:", type(exc).__name__, exc)
    return t, state, g


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


if mode == "small":
    for config, kw in (("nuscenes_gs25600_solid", dict(P=300, H=24, W=20, D=16)), ("nuscenes_gs144000", dict(P=1000, H=40, W=44, D=16)),
                       ("nuscenes_gs25600_solid", dict(P=257, H=23, W=21, D=16)), ("nuscenes_gs25600_solid", dict(P=200, H=20, W=20, D=10)),
                       ("nuscenes_gs25600_solid", dict(P=0, H=12, W=12, D=8)), ("nuscenes_gs144000", dict(P=6000, H=20, W=20, D=16))):
        run(make_splat_inputs(config, seed=3, **kw), f"{config} {kw}")
    run(make_splat_inputs("nuscenes_gs25600_solid", seed=5, P=400, H=24, W=24, D=16), "tiny gradients (x 1e-6)", grad_scale=1e-6)
    run(make_splat_inputs("nuscenes_gs25600_solid", seed=5, P=400, H=24, W=24, D=16), "huge gradients (x 1e6)", grad_scale=1e6)
else:
    for config in configs:
        si = make_splat_inputs(config, seed=0)
        t, state, g = run(si, config)
        for name, flags in (("exact (Gaussian-major)", _lib.GF_EXACT_FP32), ("auto (both pipelines gated)", 0), ("matrix cores asserted", _lib.GF_MFMA_SPLAT)):
            us = timed(lambda: splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=flags))
            print(f"{config}: backward {name}: {us:.1f} us per call (module-level: allocations included)", flush=True)
        for name, fl in (("plain", 0), ("GF_PREPARE_BACKWARD", _lib.GF_PREPARE_BACKWARD)):
            us = timed(lambda: splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=fl), n=200)
            print(f"{config}: forward {name}: {us:.1f} us per call (module-level: allocations included)", flush=True)
        # a fresh forward: the workspace holds its records, and the backward calls below leave them alone
        logits, _, _, _, state2 = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
        ref_out = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=_lib.GF_MFMA_SPLAT)
        logits, _, _, _, state2 = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
        for name, flags in (("auto, forward's records still there", 0), ("matrix cores asserted, forward's records still there", _lib.GF_MFMA_SPLAT),
                            ("matrix cores + records asserted (GF_RECORDS_VALID)", _lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID)):
            us = timed(lambda: splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state2, flags=flags))
            out = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state2, flags=flags)
            same = all(bool(torch.equal(a, b)) for a, b in zip(out, ref_out))
            print(f"{config}: backward {name}: {us:.1f} us per call; equal to the re-run records pass's result: {same}", flush=True)
