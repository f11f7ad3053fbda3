"""Development probe (GPU box): one splat forward between other work -- after a 512 MB fill that evicts L2 and the Infinity
Cache, and after an idle gap -- against the back-to-back figure, for both render kernels.  Event-timed per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev

dev = torch.device("cuda:0")
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = prep(si)
t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
junk = torch.empty(128 << 20, device=dev)
for name, flags in (("exact", _lib.GF_EXACT_FP32), ("mfma", 0)):
    plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=flags)
    for _ in range(50):
        plan.run()
    torch.cuda.synchronize()
    for mode in ("back to back", "after a 512 MB fill", "after 5 ms idle", "after fill + idle"):
        ts = []
        for i in range(30):
            if "fill" in mode:
                junk.fill_(float(i))
            if "idle" in mode:
                torch.cuda.synchronize()
                time.sleep(0.005)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan.run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[5:])
        print(f"{config} {name:5s} {mode:22s}: median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f}  max {ts[-1]:6.1f}", flush=True)
