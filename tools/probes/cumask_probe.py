#!/usr/bin/env python
"""VERDICT r3 item 1(a), the probe: does PARTITIONING the chip beat free co-scheduling of an HBM-bound and an MFMA-bound kernel?

corun_probe.py (round 2) showed that a 3.3 GB device copy (A, HBM-bound) and the 1024 -> 256 GEMM (B, MFMA-bound) launched on two
ordinary streams overlap partially (44 % of A hides under B).  Here A runs on a stream confined to the first S CUs of the mask
order (mm_stream_create_cu_mask; bits go round-robin over the 8 XCDs, so S = 64 is 8 CUs on every XCD) and B on the stream that
owns the complementary 256 - S CUs.  Reported per S: A alone on its partition (GB/s), B alone on the complement (TFLOP/s), the
pair co-running (ms for N of each), against the free co-run and the serial sum.  A second table uses the pipeline's real movement
kernel (the F(4x4,3x3) Winograd input transform through mm_resnet50-sized planes is not exported, so the max-pool-sized torch copy
stands in for it) -- the question is only how many CUs an HBM-bound stream needs.

gpurun -- 'python tools/probes/cumask_probe.py > gpurun_out/cumask_probe.txt'"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
NCU = torch.cuda.get_device_properties(0).multi_processor_count
WORDS = (NCU + 31) // 32


def masked_stream(bits):
    m = (ctypes.c_uint32 * WORDS)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    rc = L.mm_stream_create_cu_mask(ctypes.byref(h), m, WORDS)
    assert rc == 0, rc
    back = (ctypes.c_uint32 * WORDS)()
    assert L.mm_stream_get_cu_mask(h, back, WORDS) == 0
    return torch.cuda.ExternalStream(h.value, device=dev), [hex(x) for x in back]


src = torch.rand(3300 * 1024 * 1024 // 8, device=dev)       # 1.65 GB read + 1.65 GB written per copy
dst = torch.empty_like(src)
B, H, W, Ci, Co = 2048, 14, 14, 1024, 256
x = torch.rand(B, H, W, Ci, device=dev) - 0.5
w = (torch.rand(Co, Ci, device=dev) - 0.5) / 32
bias = torch.rand(Co, device=dev)
out = torch.empty(B, H, W, Co, device=dev)
GB = src.numel() * 4 * 2 / 1e9
FL = 2.0 * B * H * W * Ci * Co


def A():
    dst.copy_(src)


def Bk():
    rc = L.mm_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), None, None, None, _lib.ptr(out), B, H, W, Ci, Ci, 0, Co, Co, 0, Co,
                          1, 1, 1, 0, 1, 0, 0, _lib.current_stream())
    assert rc == 0


N = 8


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def on(stream, fn, n=N):
    def run():
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            for _ in range(n):
                fn()
        cur.wait_stream(stream)
    return run


def pair(sa, sb):
    def run():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            for _ in range(N):
                A()
        with torch.cuda.stream(sb):
            for _ in range(N):
                Bk()
        cur.wait_stream(sa); cur.wait_stream(sb)
    return run


for _ in range(2):
    A(); Bk()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ta = timed(on(s1, A)); tb = timed(on(s2, Bk)); tfree = timed(pair(s1, s2))
print("device: %d CUs, mask words %d" % (NCU, WORDS))
print("unmasked: A alone %.3f ms (%.0f GB/s)  B alone %.3f ms (%.1f TFLOP/s)  free co-run of %d+%d: %.2f ms  (serial %.2f, max %.2f)"
      % (ta / N, GB / (ta / N) * 1e3, tb / N, FL / (tb / N) / 1e9, N, N, tfree, ta + tb, max(ta, tb)))
print("%-28s %10s %10s %10s %10s %10s" % ("partition (A | B)", "A GB/s", "B TFLOP/s", "pair ms", "vs free", "vs serial"))
for kind in ("prefix", "xcd"):
    for S in (16, 32, 48, 64, 96, 128):
        if kind == "prefix":
            a_bits = list(range(S))
        else:
            # whole XCDs: bit b belongs to XCD b % 8 (if the round-robin reading of the mask is right): S/32 XCDs for A
            nx = S // 32
            if nx < 1:
                continue
            a_bits = [b for b in range(NCU) if b % 8 < nx]
        b_bits = [b for b in range(NCU) if b not in set(a_bits)]
        sa, ma = masked_stream(a_bits)
        sb, mb = masked_stream(b_bits)
        t_a = timed(on(sa, A)); t_b = timed(on(sb, Bk)); t_p = timed(pair(sa, sb))
        print("%-28s %10.0f %10.1f %10.2f %10.3f %10.3f" % ("%s %d | %d" % (kind, len(a_bits), len(b_bits)), GB / (t_a / N) * 1e3,
                                                              FL / (t_b / N) / 1e9, t_p, t_p / tfree, t_p / (ta + tb)))
        del sa, sb
# B on a masked stream of ALL CUs (does the mask path itself cost anything?)
sall, _ = masked_stream(list(range(NCU)))
print("B on a full-mask stream: %.3f ms (unmasked %.3f)" % (timed(on(sall, Bk)) / N, tb / N))
