#!/usr/bin/env python
"""Does data a kernel has just WRITTEN stay in the 256 MB Infinity Cache for the next kernel's reads?
write S bytes (fill_), read them back (sum), event-timed; S from 16 MB to 2 GB.  gpurun -- 'python tools/probes/mall_probe.py'"""
import torch

dev = torch.device("cuda:0")
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    buf = torch.empty(n, device=dev)
    other = torch.empty(512 * 1024 * 1024 // 4, device=dev)
    res = {}
    for mode in ("after write", "after flush"):
        ts = []
        for rep in range(6):
            buf.fill_(float(rep))
            if mode == "after flush":
                other.fill_(1.0)                 # 512 MB of other traffic between the write and the read
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s = buf.sum()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        res[mode] = mb / 1024 / (t * 1e-3)
    print("%5d MB: read %7.0f GB/s right after the write, %7.0f GB/s after 512 MB of other writes" % (mb, res["after write"] * 1.0737, res["after flush"] * 1.0737))
