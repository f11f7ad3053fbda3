"""Development probe (GPU box): does the deformable-aggregation forward get faster per visited camera when all points
sample ONE camera (one camera's pyramid = 14.7 MB as the L2 working set) instead of three random ones?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussianformer_amd.synthetic import make_daf_inputs
from gaussianformer_amd.deformable_aggregation import deformable_aggregation_forward as fwd
dev = torch.device("cuda:0")
d = make_daf_inputs(num_pts=230400, seed=0)
def run(loc, label):
    t = [torch.from_numpy(a).to(dev) for a in (d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"], loc, d["weights"])]
    for _ in range(5): fwd(*t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): fwd(*t)
    e1.record(); torch.cuda.synchronize()
    vis = ((loc > 0) & (loc < 1)).all(-1).sum() / loc.shape[1]
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{label}: {us:.0f} us, {vis:.2f} visible cameras per point -> {us / vis:.0f} us per camera visit", flush=True)
loc = d["sampling_location"]
run(loc, "random cameras (op-level distribution)")
one = np.full_like(loc, -1.0); one[:, :, 0] = np.clip(loc[:, :, 0], 0.01, 0.99)
run(one, "camera 0 only, every point")
two = np.full_like(loc, -1.0); two[:, :, 0] = np.clip(loc[:, :, 0], 0.01, 0.99); two[:, :, 3] = np.clip(loc[:, :, 3], 0.01, 0.99)
run(two, "cameras 0 and 3, every point")
srt = one.copy(); order = np.lexsort((srt[0, :, 0, 0], (srt[0, :, 0, 1] * 27).astype(int)))
srt = np.ascontiguousarray(srt[:, order])
run(srt, "camera 0 only, points sorted by image row band")
