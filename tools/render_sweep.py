"""Development: render-kernel time (hipEvents on the launch stream) for the bench workload(s) under the current GF_RENDER_WAVES."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
dev = torch.device("cuda:0")
lib = _lib.load()
for config in sys.argv[1:] or ["nuscenes_gs25600_solid", "nuscenes_gs144000"]:
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier,
                                                      radii_min=1 if si.variant == "prob" else None)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
    plan = SplatForwardPlan(variant, *t, si.H, si.W, si.D, flags=0)
    for _ in range(10): plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): plan.run()
    e1.record(); torch.cuda.synchronize()
    step = e0.elapsed_time(e1) / 200 * 1e3
    lib.gf_profile_stride(1); lib.gf_profile_enable(100)
    for _ in range(100): plan.run()
    torch.cuda.synchronize()
    buf = (ctypes.c_float * 100)(); n = lib.gf_profile_read(buf, 100); lib.gf_profile_enable(0)
    print(f"GF_RENDER_WAVES={os.environ.get('GF_RENDER_WAVES','default')} {config}: step {step:.1f} us, render kernel {np.mean(buf[:n])*1e3:.1f} us (min {np.min(buf[:n])*1e3:.1f})", flush=True)
