"""Development probe (GPU box): step time of the default splat forward, median / min / max of R repeats of K back-to-back steps.
GF_LIB=<path> selects a library variant.  python tools/fwd_time.py [config ...] [--reps R] [--steps K] [--flags F]"""
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

args = sys.argv[1:]
def opt(name, default):
    if name in args:
        i = args.index(name)
        v = args[i + 1]
        del args[i:i + 2]
        return int(v)
    return default
reps, steps, flags = opt("--reps", 7), opt("--steps", 200), opt("--flags", 0)
if "--tile" in args:   # the matrix-core forward on the tile kernel everywhere (library option)
    args.remove("--tile")
    _lib.set_option("splat.mfma_tile_kernel", 1)
dev = torch.device("cuda:0")
configs = args or ["nuscenes_gs25600_solid", "nuscenes_gs144000"]
for config in configs:
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=flags)
    out = plan.run().clone()
    torch.cuda.synchronize()
    for _ in range(50):
        plan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e6)
    ts.sort()
    print(f"{os.path.basename(_lib.LIB_PATH)} {config} flags={flags}: median {ts[len(ts) // 2]:.2f} us  min {ts[0]:.2f}  max {ts[-1]:.2f}  "
          f"state {plan.state_words()[:3]}  checksum {float(out.double().sum()):.6f} absmax {float(out.abs().max()):.4f}", flush=True)
