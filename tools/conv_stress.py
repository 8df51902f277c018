"""Determinism / race stress of the conv engine: many launches of the same problem must be bit-identical
(the engine's cross-wave LDS hand-offs are timing sensitive; this caught a missing lgkmcnt(0) at the barrier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()


def stress(B, H, Ci, Co, k, tile, reps=30, korder=0):
    x = torch.rand(B, H, H, Ci, device=dev) - 0.5
    kp = (k * k * Ci + 15) // 16 * 16
    w = torch.zeros(Co, kp, device=dev)
    w[:, : k * k * Ci] = torch.rand(Co, k * k * Ci, device=dev) - 0.5
    ref, bad, words = None, 0, 0
    for _ in range(reps):
        out = torch.empty(B, H, H, Co, device=dev)
        rc = L.mm_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), None, None, None, None, _lib.ptr(out), B, H, H, Ci, Ci, 0, Co, Co, 0, Co,
                              k, k, 1, k // 2, 0, tile, korder, _lib.current_stream())
        assert rc == 0
        if ref is None:
            ref = out
        else:
            n = int((out != ref).sum())
            bad += n > 0
            words += n
    print("B%d %dx%d Ci%d Co%d k%d tile%d: runs differing %d/%d (words %d)" % (B, H, H, Ci, Co, k, tile, bad, reps - 1, words), flush=True)
    return bad


if __name__ == "__main__":
    total = 0
    for tile in (3, 2, 1):
        total += stress(1600000, 1, 64, 64, 1, tile, 12)
        total += stress(400000, 1, 16, 64, 1, tile, 20)
        total += stress(100000, 1, 256, 128, 1, tile, 20)
        total += stress(512, 28, 64, 64, 3, tile, 12, 1)      # (tiles 2 / 1: the nine-tap unrolled loop, KMODE 11)
    total += stress(1024, 48, 24, 64, 3, 2, 12, 0)             # the fourteen-chunk unrolled loop (KMODE 10)
    total += stress(2048, 24, 128, 128, 3, 1, 12, 1)           # KMODE 11 on 128x128, eight slices
    print("TOTAL differing runs:", total)
    sys.exit(1 if total else 0)
