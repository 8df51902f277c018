#!/usr/bin/env python
"""Micro-benchmark of the conv/GEMM engine through the C ABI (mm_conv2d_nhwc) on one shape.
usage: conv_bench.py B H W Cin Cout k stride pad [tile] [iters] [res] [korder]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import _lib  # noqa: E402


def run(B, H, W, Ci, Co, k, st, pad, tile=0, iters=20, res=0, korder=0, relu=1):
    dev = torch.device("cuda:0")
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    K = k * k * Ci
    Kp = (K + 15) // 16 * 16
    x = torch.rand(B, H, W, Ci, device=dev) - 0.5
    w = (torch.rand(Co, Kp, device=dev) - 0.5) / K ** 0.5
    b = torch.rand(Co, device=dev)
    r = torch.rand(B, Ho, Wo, Co, device=dev) if res else None
    out = torch.empty(B, Ho, Wo, Co, device=dev)
    L = _lib.lib()

    def go():
        rc = L.mm_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r), None, None, _lib.ptr(out),
                              B, H, W, Ci, Ci, 0, Co, Co, 0, Co, k, k, st, pad, relu, tile, korder, _lib.current_stream())
        assert rc == 0, rc
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * Ho * Wo * K * Co
    print("B%d %dx%d Cin%d Cout%d k%d s%d tile%d res%d korder%d: M=%d K=%d  %.3f ms  %.1f TFLOP/s" %
          (B, H, W, Ci, Co, k, st, tile, res, korder, B * Ho * Wo, K, ms, fl / ms / 1e9))
    return ms


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    if len(a) >= 8:
        run(*a)
    else:
        run(1, 64, 64, 4096, 4096, 1, 1, 0, 1)       # 4096^3 GEMM
        for ko in (0, 1):
            run(512, 14, 14, 256, 256, 3, 1, 1, 1, 20, 0, ko)       # ResNet conv4_x 3x3
            run(512, 28, 28, 128, 128, 3, 1, 1, 1, 20, 0, ko)       # conv3_x 3x3
            run(512, 7, 7, 512, 512, 3, 1, 1, 1, 20, 0, ko)         # conv5_x 3x3
            run(512, 56, 56, 64, 64, 3, 1, 1, 2, 20, 0, ko)         # conv2_x 3x3
        run(512, 14, 14, 1024, 256, 1, 1, 0, 1)
        run(512, 14, 14, 256, 1024, 1, 1, 0, 1, 20, 1)
