import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "./tests")
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
dev = torch.device("cuda:0")
for config in ["nuscenes_gs25600_solid", "nuscenes_gs144000"]:
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    for name, flags in (("auto", 0), ("assume_dense", _lib.GF_PTS_ASSUME_DENSE)):
        plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=flags)
        for _ in range(20): plan.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): plan.run()
        torch.cuda.synchronize()
        print(f"{config} {name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step", flush=True)
