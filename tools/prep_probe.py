"""Development probe (GPU box, under rocprofv3 --kernel-trace --stats): the prep launch with and without its verification waves,
and the verification waves nearly alone (64 Gaussians)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
dev = torch.device("cuda:0")
for P, flags, tag in ((None, 0, "auto"), (None, _lib.GF_PTS_ASSUME_DENSE, "assume_dense"), (63, 0, "auto_P64")):
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=0) if P is None else make_splat_inputs("nuscenes_gs25600_solid", seed=0, P=P)
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    plan = SplatForwardPlan(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=flags)
    for _ in range(300):
        plan.run()
    torch.cuda.synchronize()
    print(tag, "done", flush=True)
    # marker kernel between phases so the trace can be split: a distinct torch op
    torch.zeros(1234567, device=dev).sum().item()
