"""Runs the splat forward (told of the backward) + matrix-core backward as a pair, as the autograd module does (for rocprofv3 runs and
event timings).  usage: prof_fb.py [config] [iters]"""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs144000"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = prep(si)
t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
N = si.pts.shape[0]
g = torch.randn(N, 18, generator=torch.Generator().manual_seed(1)).to(dev)
BW = _lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf, tb = [], []
for i in range(iters + 3):
    ev[0].record()
    lg, _, _, _, st = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
    ev[1].record()
    out = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=st, flags=BW)
    ev[2].record()
    torch.cuda.synchronize()
    if i >= 3:
        tf.append(ev[0].elapsed_time(ev[1]) * 1e3); tb.append(ev[1].elapsed_time(ev[2]) * 1e3)
print(config, "state", st.view(torch.int32)[:5].tolist(), f"forward(prepared) {np.median(tf):.1f} us, backward {np.median(tb):.1f} us (events, one pair per sync)")
ex = splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=st, flags=_lib.GF_EXACT_FP32)
for name, a, b in zip(("means", "opacity", "semantics", "cov"), out, ex):
    d = (a - b).abs().max().item(); m = b.abs().max().item()
    print(f"  {name}: max|diff| {d:.3e} of max|exact| {m:.3e}")
