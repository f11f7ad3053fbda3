"""Microbenchmark (GPU box): does a buffer that is rewritten in place come back from the Infinity Cache (256 MB)?  x.mul_(c) -- one read and
one write of every byte -- repeated on buffers of growing size; bytes moved per second by the events around 20 repeats.
python tools/microbench/mall_probe.py"""
import torch
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 3072):
    n = mb * 1024 * 1024 // 4
    x = torch.ones(n, device=dev)
    for _ in range(3): x.mul_(1.0001)
    torch.cuda.synchronize()
    reps = 20
    e0.record()
    for _ in range(reps): x.mul_(1.0001)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mb:5d} MB in place: {ms * 1e3:8.1f} us per pass = {2 * mb / 1024 / 1.024 / ms:6.2f} TB/s (read + write)", flush=True)
