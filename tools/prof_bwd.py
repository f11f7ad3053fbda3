"""Runs the splat forward + backward a few times (for rocprofv3 runs).  usage: prof_bwd.py [config] [iters] [bwd flags]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle
config = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_gs25600_solid"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
si = make_splat_inputs(config, seed=0)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size,
                                                  si.scale_multiplier, radii_min=1 if si.variant == "prob" else None)
t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
logits, bl, de, pr, state = splat_forward(variant, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD if flags & _lib.GF_RECORDS_VALID else 0)
g = torch.randn(logits.shape, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(iters):
    splat_backward(variant, *t, si.H, si.W, si.D, g, fwd_outputs=(logits, bl, de, pr) if variant else None, state=state, flags=flags)
torch.cuda.synchronize()
