"""Round 6: phase_window2_kernel_pair (both levels of a (window, band) in one twelve-wave workgroup) against one launch per level
(MM_PW_PAIR=0): outputs compared value for value in both layouts, window-kernel time of both through the library's hipEvent hook.
usage: python tools/pw_pair_check.py [clips ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib, synthetic, sampler
from mimamo_net_amd.phase_difference_extractor import Phase_Difference_Extractor

dev = torch.device("cuda:0")
pde = Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
L = _lib.lib()


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 5)(); work = (ctypes.c_double * 5)(); launches = (ctypes.c_int64 * 5)()
    L.mm_profile_begin()
    for _ in range(reps):
        f()
    L.mm_profile_end(ms, work, launches)
    return ms[1] / reps, ms[2] / reps, launches[2] // reps


for clips in [int(a) for a in sys.argv[1:]] or [1, 3, 11, 32, 64]:
    n = clips * 64
    base = np.concatenate([synthetic.preprocess_host(synthetic.make_clip_u8(c, 64))[0] for c in range(min(clips, 32))])
    gray = torch.from_numpy(base).to(dev).repeat((clips + 31) // 32, 1, 1)[:n].contiguous()
    if clips == 3:
        gray[64:128] = gray[64:65]          # a still clip: no wrap at all
        gray[130] *= 1e4
    ids = torch.from_numpy(np.concatenate([sampler.window_ids(0, 64, 64) + 64 * c for c in range(clips)]).astype(np.int32)).to(dev)
    res = {}
    with torch.no_grad():
        for mode in ("0", "1"):
            os.environ["MM_PW_PAIR"] = mode
            a = pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64, ids_checked=True)
            b = pde.phase_diff_frames(gray, ids, ids_checked=True)
            torch.cuda.synchronize()
            outs = [a[0].clone(), a[1][..., 64:].clone(), b[0].clone(), b[1].clone()]
            t = timed(lambda: pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64, ids_checked=True))
            res[mode] = (outs, t)
    same = [torch.equal(x, y) for x, y in zip(res["0"][0], res["1"][0])]
    worst = max((x - y).abs().max().item() for x, y in zip(res["0"][0], res["1"][0]))
    fin = all(torch.isfinite(x).all().item() for x in res["1"][0])
    t0, t1 = res["0"][1], res["1"][1]
    print("%3d clips (%5d frames): window kernels %.4f ms in %d launches -> %.4f ms in %d (x%.2f); frame kernel %.4f / %.4f ms | NHWC L1 %s L2 %s, NCHW L1 %s L2 %s, "
          "max |diff| %.3g, finite %s" % (clips, n, t0[1], t0[2], t1[1], t1[2], t0[1] / t1[1], t0[0], t1[0], *("equal" if s_ else "DIFF" for s_ in same), worst, fin), flush=True)
