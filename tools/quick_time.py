"""Quick device timing of the splat forward (hipEvents via torch) -- development aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_forward, splat_backward
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle

dev = torch.device("cuda:0")
for config in sys.argv[1:] or ["nuscenes_gs25600_solid", "nuscenes_gs144000"]:
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size,
                                                      si.scale_multiplier, radii_min=1 if si.variant == "prob" else None)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
         (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
    ref = None
    for name, flags in (("auto", 0), ("auto+comp_exp", 16), ("auto+libm_exp", 8), ("assume_dense", 1), ("general", 2)):
        for _ in range(5):
            out = splat_forward(variant, *t, si.H, si.W, si.D, flags=flags)
        torch.cuda.synchronize()
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = splat_forward(variant, *t, si.H, si.W, si.D, flags=flags)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        P = si.means3D.shape[0]
        if ref is None:
            ref = oracle.splat_forward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                       si.H, si.W, si.D)["logits"]
        err = np.abs(out[0].cpu().numpy() - ref) / np.maximum(1.0, np.abs(ref))
        print(f"{config} fwd[{name}]: {ms*1e3:.1f} us/call  {P/ms/1e6:.3f} G Gaussians/s   max scaled err vs oracle {err.max():.2e}", flush=True)
    g = torch.randn(si.pts.shape[0], 18, device=dev)
    logits, bl, de, pr, state = splat_forward(variant, *t, si.H, si.W, si.D)
    for _ in range(3):
        splat_backward(variant, *t, si.H, si.W, si.D, g, fwd_outputs=(logits, bl, de, pr) if variant else None, state=state)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        splat_backward(variant, *t, si.H, si.W, si.D, g, fwd_outputs=(logits, bl, de, pr) if variant else None, state=state)
    e1.record(); torch.cuda.synchronize()
    print(f"{config} bwd: {e0.elapsed_time(e1)/20*1e3:.1f} us/call", flush=True)
