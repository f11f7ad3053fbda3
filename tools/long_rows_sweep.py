"""Development probe (GPU box): seeded sweep of the long-row kernels (39 552 < P <= 262 144) -- forward wave kernel against the tile
kernel (equal bits), matrix-core backward (prepared by the forward) against the unprepared one (equal bits where the rows fit) and
against the exact Gaussian-major kernels (row by row, 1e-3).  python tools/long_rows_sweep.py [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev, grad_row_errors, whole_grid_rows
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(606)
bad = 0
for it in range(n):
    P = int(rng.integers(39553, 150000)) if it % 4 else int(rng.integers(150000, 262144))
    H, W, D = int(rng.integers(16, 120)), int(rng.integers(16, 120)), int(rng.choice([8, 16, 24]))
    config = "nuscenes_gs144000" if it % 3 else "nuscenes_gs25600_solid"
    si = make_splat_inputs(config, seed=700 + it, P=P, H=H, W=W, D=D, clustered=bool(it % 2))
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    g = torch.randn(si.pts.shape[0], 18, generator=torch.Generator().manual_seed(it)).to(dev)
    whole = whole_grid_rows(mi, radii, si.H, si.W, si.D)
    lg0, _, _, _, st0 = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
    lg0 = lg0.clone()
    plain = [x.clone() for x in splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=st0)]
    exact = [x.clone() for x in splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=st0, flags=_lib.GF_EXACT_FP32)]
    with _lib.option("splat.mfma_tile_kernel", 1):
        lgt, _, _, _, _ = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
        lgt = lgt.clone()
    lg, _, _, _, st = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
    torch.cuda.synchronize()
    words = st.view(torch.int32)[:5].tolist()
    prepared = bool(words[4] & 1)
    got = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=st, flags=(_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if prepared else 0)
    torch.cuda.synchronize()
    fwd_equal = bool(torch.equal(lg0, lgt)) and bool(torch.equal(lg, lg0))
    bwd_equal = all(bool(torch.equal(a, b)) for a, b in zip(got, plain)) if prepared else None
    errs = [grad_row_errors(a.cpu().numpy(), b.cpu().numpy(), whole) for a, b in zip(got, exact)]
    worst = max(max(e["ordinary"], e["whole_grid"]) for e in errs)
    ok = fwd_equal and (bwd_equal in (True, None)) and worst <= 1e-3 and all(bool(torch.isfinite(x).all()) for x in got)
    bad += 0 if ok else 1
    print(f"{it}: {config} clustered={bool(it % 2)} P={P} {H}x{W}x{D} state {words}: forward wave == tile: {fwd_equal}; backward prepared == unprepared: {bwd_equal}; "
          f"worst row error against the exact kernels {worst:.2e} {'OK' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
