"""Development probe (GPU box): gf_daf_fused_forward against the three-step path at 25 600 and 144 000 anchors (projected geometry, the
frame benchmark's cameras) -- time and row-scaled difference.  GF_LIB selects a library variant.  python tools/daf_fused_time.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_frame
from gaussianformer_amd import _lib
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_forward
from gaussianformer_amd.deformable_prepare import deformable_fused_forward, deformable_prepare
from gaussianformer_amd.synthetic import make_daf_inputs
dev = torch.device("cuda:0")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for A in (25600, 144000):
    g = torch.Generator(device="cpu").manual_seed(1)
    lo = torch.tensor(bench_frame.PC_RANGE[:3]); hi = torch.tensor(bench_frame.PC_RANGE[3:])
    centre = lo + (hi - lo) * torch.rand(1, A, 3, generator=g)
    offs = torch.tensor(bench_frame.FIX_SCALE + [[0.3, 0.3, 0.0], [-0.3, 0.3, 0.0]]) * 0.35
    kp = (centre[:, :, None] + offs[None, None]).to(dev)
    pm, wh = bench_frame.cameras(dev)
    ra = torch.randn(1, A, 4, 9, 4, generator=g).to(dev)
    rc = torch.randn(1, 6, 4, 9, 4, generator=g).to(dev)
    d = make_daf_inputs(num_pts=9, seed=0)
    feat, ss, st = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index"))
    with torch.no_grad():
        raw = (ra[:, :, None] + rc[:, None]).reshape(1, A, 6, 4, 9, 4)
        loc, w = deformable_prepare(kp, pm, wh, raw)
        want = deformable_aggregation_forward(feat, ss, st, loc, w).reshape(1, A, 9, 128).sum(dim=2)
        got = deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_anchor=ra, raw_cam=rc)
        err = float(((got - want).abs() / want.abs().amax(dim=-1, keepdim=True).clamp(min=1e-3)).max())
        t = timed(lambda: deformable_fused_forward(kp, pm, wh, feat, ss, st, raw_anchor=ra, raw_cam=rc))
    print(f"{os.path.basename(_lib.LIB_PATH)} A={A}: fused {t:.1f} us, max row-scaled difference from the three-step path {err:.2e}", flush=True)
