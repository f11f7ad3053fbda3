"""Development probe (GPU box): DAF forward, plain walk (development build, option "dev.daf_plain") against the batched-loads kernel, both location distributions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_forward
from gaussianformer_amd.synthetic import make_daf_inputs
dev = torch.device("cuda:0")
def timed(fn, warm=5, iters=40):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
pts = 230400
d = make_daf_inputs(num_pts=pts, seed=0)
feat, ss, st, loc, w = (torch.from_numpy(d[k]).to(dev) for k in ("mc_ms_feat", "spatial_shape", "scale_start_index", "sampling_location", "weights"))
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
ploc, pw = deformable_prepare(kp, pm, wh, raw)
ploc, pw = ploc.contiguous(), pw.contiguous()
for name, (l_, w_) in (("uniform", (loc, w)), ("projected", (ploc, pw))):
    outs = {}
    for mode in ("plain", "batched"):
        _lib.set_option("dev.daf_plain", 1 if mode == "plain" else 0)   # (development build: GF_LIB=.../libgf_hip_dev.so)
        outs[mode] = deformable_aggregation_forward(feat, ss, st, l_, w_).clone()
        print(f"{name:10s} {mode:8s}: {timed(lambda: deformable_aggregation_forward(feat, ss, st, l_, w_)):7.1f} us", flush=True)
    print(f"{name:10s} bit-identical: {bool(torch.equal(outs['plain'], outs['batched']))}")
