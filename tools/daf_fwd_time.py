"""Development probe (GPU box): deformable aggregation forward at 230 400 points, projected and uniform locations -- time and a
checksum (library variants must give the same bits).  GF_LIB selects the variant.  python tools/daf_fwd_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_forward
from gaussianformer_amd.synthetic import make_daf_inputs
dev = torch.device("cuda:0")
pts = 230400
for dist in (sys.argv[1:] or ["projected", "uniform"]):
    d = make_daf_inputs(num_pts=pts, seed=0)
    feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
    if dist == "projected":
        import bench_frame
        from gaussianformer_amd.deformable_prepare import deformable_prepare
        g = torch.Generator(device="cpu").manual_seed(1)
        A = pts // 9
        lo = torch.tensor(bench_frame.PC_RANGE[:3]); hi = torch.tensor(bench_frame.PC_RANGE[3:])
        centre = lo + (hi - lo) * torch.rand(1, A, 3, generator=g)
        offs = torch.tensor(bench_frame.FIX_SCALE + [[0.3, 0.3, 0.0], [-0.3, 0.3, 0.0]]) * 0.35
        kp = (centre[:, :, None] + offs[None, None]).to(dev)
        pm, wh = bench_frame.cameras(dev)
        raw = torch.randn(1, A, 6, 4, 9, 4, generator=g).to(dev)
        loc, w = deformable_prepare(kp, pm, wh, raw)
        loc, w = loc.contiguous(), w.contiguous()
    y = deformable_aggregation_forward(feat, ss, st, loc, w)
    torch.cuda.synchronize()
    for _ in range(5):
        deformable_aggregation_forward(feat, ss, st, loc, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        deformable_aggregation_forward(feat, ss, st, loc, w)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(_lib.LIB_PATH)} {dist}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per call, checksum {float(y.double().sum()):.9e} absmax {float(y.abs().max()):.6f}", flush=True)
