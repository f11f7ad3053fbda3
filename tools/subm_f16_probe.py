"""Development probe (GPU box): gather-GEMM organisations of the sparse convolution side by side -- two f16 terms (default), three bf16
terms ("subm.bf16x3"), f32 MFMA ("subm.f32_mfma") -- apply time and error against the f32-MFMA kernel.  python tools/subm_f16_probe.py [A]"""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianformer_amd import _lib
from gaussianformer_amd.sparse_conv import Rulebook
dev = torch.device("cuda:0")
for A in ([int(a) for a in sys.argv[1:]] or [25600, 144000]):
    torch.manual_seed(0)
    xyz = torch.rand(A, 3, device=dev) * torch.tensor([160.0, 160.0, 16.0], device=dev)
    idx = torch.cat([torch.zeros(A, 1, dtype=torch.int32, device=dev), xyz.to(torch.int32)], dim=1)
    feat = torch.randn(A, 128, device=dev)
    w = torch.randn(125, 128, 128, device=dev) * 0.05
    rb = Rulebook(idx, 1, (160, 160, 16), 5)
    def timed(n=20):
        for _ in range(3): rb.apply(feat, w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): rb.apply(feat, w)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    with _lib.option("subm.f32_mfma", 1):
        exact = rb.apply(feat, w); t_exact = timed()
    res = {}
    for name, opt in (("f16x2", None), ("bf16x3", "subm.bf16x3")):
        if opt:
            with _lib.option(opt, 1):
                out = rb.apply(feat, w); t = timed()
        else:
            out = rb.apply(feat, w); t = timed()
        res[name] = (t, float((out - exact).abs().max() / exact.abs().max()))
    print(f"A={A} pairs={rb.total}: f32 mfma {t_exact:.0f} us; " + "; ".join(f"{k} {v[0]:.0f} us (max diff / max|out| {v[1]:.2e})" for k, v in res.items()), flush=True)
