"""Development aid: does a spatial order of the Gaussians (plus the XCD-aware range schedule)
cut the backward's L2-miss traffic?  Permutes the INPUT Gaussians on the host and times
gf_splat_backward for each (order, library variant)."""
import os
import subprocess
import sys
sys.path.insert(0, ".")

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    import oracle
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
    from gaussianformer_amd.synthetic import make_splat_inputs
    dev = torch.device("cuda:0")
    for config in ("nuscenes_gs25600_solid", "nuscenes_gs144000"):
        si = make_splat_inputs(config, seed=0)
        pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size,
                                                          si.scale_multiplier, radii_min=None)
        P = mi.shape[0]
        cx, cy = np.clip(mi[:, 0], 0, si.H - 1), np.clip(mi[:, 1], 0, si.W - 1)
        orders = {
            "index": np.arange(P),
            "x": np.argsort(cx, kind="stable"),
            "cell25x50": np.argsort((cx // 25) * 4 + cy // 50, kind="stable"),
            "cell25x25": np.argsort((cx // 25) * 8 + cy // 25, kind="stable"),
            "xy": np.argsort(cx * si.W + cy, kind="stable"),
        }
        for oname, perm in orders.items():
            arrs = [si.pts, pi, si.means3D[perm], mi[perm], si.opacities[perm], si.semantics[perm], radii[perm], cov6[perm]]
            t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]
            logits, bl, de, pr, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
            g = torch.randn(pi.shape[0], 18, device=dev)
            fn = lambda: splat_backward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, g, state=state)
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"   {config:24s} order {oname:10s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
    sys.exit(0)

from gaussianformer_amd import build as B
for spec in sys.argv[1:] or ["base:"]:
    name, _, flags = spec.partition(":")
    lib = B.build(extra_flags=tuple(flags.split()), lib_name=f"libgf_hip_{name}.so")
    print(f"==== variant {name} [{flags}]", flush=True)
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, GF_LIB=lib), capture_output=True, text=True)
    print(r.stdout, end="", flush=True)
    if r.returncode:
        print("   FAILED", r.stderr[-800:], flush=True)
