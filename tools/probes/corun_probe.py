#!/usr/bin/env python
"""Do an HBM-bound kernel and a matrix-pipe-bound kernel share the chip when launched on two streams?
A = device copy of 3.3 GB (HBM-bound), B = 1x1 GEMM 401408 x 1024 x 256 on the conv engine (MFMA-bound), N of each back to back:
one stream A..A B..B, one stream interleaved ABAB.., two streams (A.. on one, B.. on the other).
gpurun -- 'python tools/probes/corun_probe.py'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
src = torch.rand(3300 * 1024 * 1024 // 8, device=dev)       # 1.65 GB read + 1.65 GB written per copy
dst = torch.empty_like(src)
B, H, W, Ci, Co = 2048, 14, 14, 1024, 256
x = torch.rand(B, H, W, Ci, device=dev) - 0.5
w = (torch.rand(Co, Ci, device=dev) - 0.5) / 32
b = torch.rand(Co, device=dev)
out = torch.empty(B, H, W, Co, device=dev)


def A():
    dst.copy_(src)


def Bk():
    rc = L.mm_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, None, None, _lib.ptr(out), B, H, W, Ci, Ci, 0, Co, Co, 0, Co,
                          1, 1, 1, 0, 1, 0, 0, _lib.current_stream())
    assert rc == 0


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


N = 8
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(N): A()
    with torch.cuda.stream(s2):
        for _ in range(N): Bk()
    cur.wait_stream(s1); cur.wait_stream(s2)


ta = timed(lambda: [A() for _ in range(N)])
tb = timed(lambda: [Bk() for _ in range(N)])
ti = timed(lambda: [(A(), Bk()) for _ in range(N)])
t2 = timed(two_streams)
print("A alone %.2f ms (%.0f GB/s)   B alone %.2f ms (%.0f TFLOP/s)" % (ta / N, 3.3 * 1.0737 / (ta / N) * 1e3, tb / N, 2.0 * B * H * W * Ci * Co / (tb / N) / 1e9))
print("one stream A..B..: %.2f ms   interleaved: %.2f ms   two streams: %.2f ms   (sum %.2f, max %.2f)" % (ta + tb, ti, t2, ta + tb, max(ta, tb)))
