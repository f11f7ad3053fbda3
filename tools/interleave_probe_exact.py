import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
dev = torch.device("cuda:0")
for config, variant, flags in (("prob_gs6400", _lib.GF_SPLAT_PROB, 0), ("nuscenes_gs25600_solid", _lib.GF_SPLAT_BASE, _lib.GF_EXACT_FP32)):
    for clustered in (False, True):
        si = make_splat_inputs(config, seed=0, clustered=clustered)
        pi, mi, radii, cov6 = prep(si)
        t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
        r = {}
        for mode in ("bands", "interleaved"):
            _lib.set_option("dev.units_bands", 1 if mode == "bands" else 0)   # (development build: GF_LIB=.../libgf_hip_dev.so)
            plan = SplatForwardPlan(variant, *t, si.H, si.W, si.D, flags=flags)
            out = plan.run().clone(); torch.cuda.synchronize()
            for _ in range(10): plan.run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): plan.run()
            torch.cuda.synchronize(); r[mode] = (out, (time.perf_counter() - t0) / 100 * 1e6)
        _lib.set_option("dev.units_bands", 0)
        print(f"{config} exact/prob tile kernel clustered={clustered}: bands {r['bands'][1]:.1f} us, interleaved {r['interleaved'][1]:.1f} us, equal bits {bool(torch.equal(r['bands'][0], r['interleaved'][0]))}", flush=True)
