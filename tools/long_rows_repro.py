"""Development probe (GPU box): the long-row wave kernel at nuscenes_gs144000 -- repeated launches equal bit for bit?  labels epilogue
equal to argmax of the logits?  the inference frame equal to itself between runs?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import splat_forward, splat_forward_labels
from gaussianformer_amd.synthetic import make_splat_inputs
from util import prep, to_dev
dev = torch.device("cuda:0")
for cfg, kw in (("nuscenes_gs144000", {}), ("nuscenes_gs144000", dict(P=72000)), ("nuscenes_gs144000", dict(P=144000, H=100))):
    si = make_splat_inputs(cfg, seed=0, **kw)
    pi, mi, radii, cov6 = prep(si)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    ref = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)[0].clone()
    bad = 0
    for i in range(40):
        out = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)[0]
        bad += int(not torch.equal(out, ref))
    lab = splat_forward_labels(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
    lab_ref = ref.argmax(dim=1)
    badl = 0
    for i in range(20):
        l2 = splat_forward_labels(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
        badl += int(not torch.equal(l2, lab))
    print(cfg, kw, "logits differ in", bad, "of 40 runs; labels differ in", badl, "of 20; labels == argmax(logits):", float((lab == lab_ref).float().mean()), flush=True)
import bench_frame
torch.manual_seed(0)
for cfg in ("nuscenes_gs144000",):
    model = bench_frame.Frame(cfg).to(dev).eval()
    A = model.cfg["anchors"]
    anchor = torch.randn(1, A, model.anchor_dim, device=dev); feat = torch.randn(1, A, bench_frame.EMBED, device=dev)
    from gaussianformer_amd.synthetic import DAF_LEVELS, voxel_centres
    maps = [torch.randn(1, bench_frame.CAMS, bench_frame.EMBED, h, w, device=dev) for h, w in DAF_LEVELS]
    pm, wh = bench_frame.cameras(dev)
    pts = torch.from_numpy(voxel_centres(200, 200, 16, 0.5, np.asarray(bench_frame.PC_RANGE[:3], dtype=np.float32))).to(dev)[None]
    want = model(anchor, feat, maps, pm, wh, pts).clone()
    fr = []
    for i in range(6):
        got = model(anchor, feat, maps, pm, wh, pts)
        fr.append(float((got == want).float().mean()))
    print(cfg, "frame labels equal to the first run:", fr, flush=True)
