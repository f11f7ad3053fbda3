import sys, os
import numpy as np, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_forward
from gaussianformer_amd.synthetic import make_splat_inputs
import oracle
from util import to_dev
dev=torch.device("cuda:0")
si = make_splat_inputs("prob_gs6400", seed=0)
print("P", si.means3D.shape, "grid", si.H, si.W, si.D, "scales min/median/max", si.scales.min(), np.median(si.scales), si.scales.max(), "grid_size", si.grid_size)
pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier, radii_min=1)
r1 = radii if radii.ndim == 1 else radii.max(axis=1)
print("radii (voxels) median", np.median(r1), "max", r1.max(), "mean box volume", float(np.mean((2*r1+1)**3)))
t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics[:, :18] if si.semantics.shape[1] >= 18 else si.semantics, np.ascontiguousarray(r1.astype(np.int32)), cov6)
print("semantics shape", si.semantics.shape)
# how many Gaussians pass the range verdict individually: run subsets
def verdict(mask):
    idx = np.flatnonzero(mask)
    tt = to_dev(dev, si.pts, pi, si.means3D[idx], mi[idx], si.opacities[idx], si.semantics[idx], np.ascontiguousarray(r1[idx].astype(np.int32)), cov6[idx])
    lg, _, _, _, st = splat_forward(_lib.GF_SPLAT_BASE, *tt, si.H, si.W, si.D, flags=_lib.GF_MFMA_SPLAT)
    torch.cuda.synchronize()
    return st.view(torch.int32)[:3].tolist()
print("all:", verdict(np.ones(len(r1), bool)))
smin = si.scales.min(axis=1)
for thr in (0.02, 0.05, 0.08, 0.1, 0.15, 0.2, 0.3):
    m = smin >= thr
    print(f"smallest scale >= {thr}: {m.mean()*100:.1f}% of the Gaussians, state", verdict(m))
