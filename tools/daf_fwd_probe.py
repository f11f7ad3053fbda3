"""Development probe (GPU box): deformable aggregation forward, eight channels per lane (default) against four (dev.daf_vec4) -- equal bits? times?"""
import os
import sys
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
    out = {}
    for mode in ("vec4", "vec8"):
        _lib.set_option("dev.daf_vec4", 1 if mode == "vec4" else 0)   # (development build: GF_LIB=.../libgf_hip_dev.so)
        y = deformable_aggregation_forward(feat, ss, st, loc, w)
        torch.cuda.synchronize()
        for _ in range(5):
            deformable_aggregation_forward(feat, ss, st, loc, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            deformable_aggregation_forward(feat, ss, st, loc, w)
        e1.record()
        torch.cuda.synchronize()
        out[mode] = (y, e0.elapsed_time(e1) / 20 * 1e3)
    _lib.set_option("dev.daf_vec4", 0)
    print(f"{dist}: four channels per lane (dev.daf_vec4) {out['vec4'][1]:.1f} us, eight (default) {out['vec8'][1]:.1f} us; equal bits {bool(torch.equal(out['vec4'][0], out['vec8'][0]))}", flush=True)
