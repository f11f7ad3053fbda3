"""Development probe (GPU box): the sparse-conv block (rulebook + gather-GEMM + output projection) on the WHOLE anchor set against
the same block evaluated for one rank's output range (1/8 of the anchors, the whole set as neighbours) -- DESIGN section 6's
cost model of the sharded frame.   python tools/subm_range_probe.py [anchors ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_frame
from gaussianformer_amd.sparse_conv import SparseConv3D

dev = torch.device("cuda:0")
world = int(os.environ.get("GF_PROBE_WORLD", "8"))
for A in [int(a) for a in sys.argv[1:]] or [25600, 144000]:
    g = torch.Generator(device="cpu").manual_seed(0)
    anchor = torch.randn(1, A, 11, generator=g).to(dev)
    feat = torch.randn(1, A, 128, generator=g).to(dev)
    grid = [0.5, 0.5, 0.5]   # (bench_frame uses the same cell for both anchor counts)
    blk = SparseConv3D(128, 128, bench_frame.PC_RANGE, grid, use_out_proj=True, kernel_size=5).to(dev).eval()
    lo, hi = 3 * (A // world), 4 * (A // world)
    out = {}
    with torch.no_grad():
        for name, kw in (("whole", {}), ("range", dict(out_range=(lo, hi)))):
            y = blk(feat, anchor, **kw)
            torch.cuda.synchronize()
            for _ in range(3):
                blk(feat, anchor, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                blk(feat, anchor, **kw)
            torch.cuda.synchronize()
            out[name] = (y, (time.perf_counter() - t0) / 20 * 1e3)
    same = bool(torch.allclose(out["whole"][0][:, lo:hi], out["range"][0], rtol=1e-5, atol=1e-5))
    pairs = getattr(blk.last_rulebook, "n_pairs", getattr(blk.last_rulebook, "pairs", -1))
    print(f"anchors {A}: whole block {out['whole'][1]:.3f} ms, range [{lo}, {hi}) of {world} ranks {out['range'][1]:.3f} ms "
          f"({out['whole'][1] / out['range'][1]:.2f}x); rows agree {same}; pairs of the range {pairs}", flush=True)
