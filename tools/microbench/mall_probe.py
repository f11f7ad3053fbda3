"""Microbenchmark (GPU box): does a buffer that was just written come back from the Infinity Cache?  Write (copy_ from a small
source is avoided: fill_) then read (sum) buffers of growing size, back to back.  python tools/microbench/mall_probe.py"""
import torch, time
dev = torch.device("cuda:0")
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 3072):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tw, tr = [], []
    for it in range(6):
        e[0].record(); x.fill_(float(it)); e[1].record(); s = x.sum(); e[2].record(); torch.cuda.synchronize()
        if it >= 2: tw.append(e[0].elapsed_time(e[1])); tr.append(e[1].elapsed_time(e[2]))
    w, r = min(tw), min(tr)
    print(f"{mb:5d} MB: write {w*1e3:8.1f} us = {mb/1024/w*1e3/1.024:6.2f} TB/s; read-after-write {r*1e3:8.1f} us = {mb/1024/r*1e3/1.024:6.2f} TB/s", flush=True)
