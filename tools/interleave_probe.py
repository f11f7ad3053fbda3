"""Development probe (GPU box): the matrix-core forward and backward with supertiles dealt round-robin to the XCDs (default since
round 5) against contiguous bands of units (development build, option "dev.units_bands"), uniform and clustered Gaussian centres -- equal bits?  times?"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev

dev = torch.device("cuda:0")
for config in (sys.argv[1:] or ["nuscenes_gs25600_solid"]):
    for clustered in (False, True):
        si = make_splat_inputs(config, seed=0, clustered=clustered)
        pi, mi, radii, cov6 = prep(si)
        t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
        res = {}
        for mode in ("bands", "interleaved"):
            _lib.set_option("dev.units_bands", 1 if mode == "bands" else 0)   # (development build: GF_LIB=.../libgf_hip_dev.so)
            plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=0)
            out = plan.run().clone()
            torch.cuda.synchronize()
            for _ in range(30):
                plan.run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(200):
                    plan.run()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 200 * 1e6)
            # backward (matrix cores, after a prepared forward), module-level calls
            from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
            g = torch.randn(si.pts.shape[0], 18, generator=torch.Generator().manual_seed(1)).to(dev)
            def fb():
                _, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
                return splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state, flags=_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID)
            grads = fb()
            torch.cuda.synchronize()
            for _ in range(10):
                fb()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                fb()
            torch.cuda.synchronize()
            fb_us = (time.perf_counter() - t0) / 100 * 1e6
            res[mode] = (out, sorted(ts)[2], plan.state_words()[:3], fb_us, [x.clone() for x in grads])
        _lib.set_option("dev.units_bands", 0)
        print(f"{config} clustered={clustered}: bands {res['bands'][1]:.2f} us {res['bands'][2]}, interleaved {res['interleaved'][1]:.2f} us {res['interleaved'][2]}; "
              f"equal bits {bool(torch.equal(res['bands'][0], res['interleaved'][0]))}; forward + backward (module calls) bands {res['bands'][3]:.1f} us, "
              f"interleaved {res['interleaved'][3]:.1f} us; gradients of all but the whole-grid Gaussian equal "
              f"{all(bool(torch.equal(a[:-1], b[:-1])) for a, b in zip(res['bands'][4], res['interleaved'][4]))}", flush=True)
