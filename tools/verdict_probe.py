import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
dev=torch.device('cuda:0')
si=make_splat_inputs("nuscenes_gs25600_solid",seed=0); pi,mi,radii,cov6=prep(si)
t=to_dev(dev,si.pts,pi,si.means3D,mi,si.opacities,si.semantics,radii,cov6)
p0=SplatForwardPlan(0,*t,si.H,si.W,si.D,flags=_lib.GF_PTS_AUTO)
a=p0.run().clone(); torch.cuda.synchronize(); print("default state",p0.state_words())
p1=SplatForwardPlan(0,*t,si.H,si.W,si.D,flags=_lib.GF_PTS_AUTO|_lib.GF_EXACT_FP32)
b=p1.run().clone(); torch.cuda.synchronize(); print("exact state",p1.state_words(), "diff", float((a-b).abs().max()))
a2=p0.run().clone(); torch.cuda.synchronize(); print("default again",p0.state_words(), float((a2-a).abs().max()), float((a2-b).abs().max()))
# graph capture of p0
g=torch.cuda.CUDAGraph()
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    p0.run(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        p0.run(s.cuda_stream)
    for _ in range(3): g.replay()
torch.cuda.synchronize()
print("after graph replays", p0.state_words(), float((p0.logits-a).abs().max()))
a3=p0.run().clone(); torch.cuda.synchronize(); print("eager after graph",p0.state_words(), float((a3-a).abs().max()))
