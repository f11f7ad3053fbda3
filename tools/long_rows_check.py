import sys, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from gaussianformer_amd import _lib
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, hip_splat_forward
gpu=torch.device('cuda:0')
cases=[("nuscenes_gs144000", {}),("nuscenes_gs144000", dict(P=50000, H=24, W=24, D=16)),("nuscenes_gs25600_solid", dict(P=45000, H=9, W=7, D=4)),
       ("nuscenes_gs144000", dict(P=40000, H=64, W=40, D=8)),("nuscenes_gs144000", dict(P=262144, H=48, W=40, D=16)),("nuscenes_gs144000", dict(P=100000,H=200,W=200,D=16))]
for cfg,kw in cases:
    si=make_splat_inputs(cfg,seed=4,**kw); pi,mi,radii,cov6=prep(si)
    w,_,ws,_=hip_splat_forward(gpu,si,pi,mi,radii,cov6,flags=_lib.GF_MFMA_SPLAT)
    with _lib.option("splat.mfma_tile_kernel",1):
        t,_,ts,_=hip_splat_forward(gpu,si,pi,mi,radii,cov6,flags=_lib.GF_MFMA_SPLAT)
    print(cfg,kw,"paths",ws[:12].view(torch.int32).tolist()[:3],ts[:12].view(torch.int32).tolist()[:3],"equal",np.array_equal(w["logits"],t["logits"]),"maxdiff",np.abs(w["logits"]-t["logits"]).max(), "finite", np.isfinite(w["logits"]).all(),flush=True)
