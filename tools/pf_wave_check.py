"""Round 6: pyramid_wave_kernel (one wave per frame, pyramid_wave.hip) against the round-3 pyramid_frame_kernel (MM_PF_WAVE=0):
the four planes per (frame, band, level) left in the workspace compared bit for bit, and both kernels timed through the library's
hipEvent hook.  usage: python tools/pf_wave_check.py [clips ...]"""
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


def planes(gray, ids):
    p0, p1 = pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64, ids_checked=True)
    torch.cuda.synchronize()
    ws = pde._ws[torch.cuda.current_stream().cuda_stream]
    n = gray.shape[0]
    a = ws[: n * 2 * 4 * 48 * 48].view(n, 2, 4, 48, 48).clone()
    b = ws[n * 2 * 4 * 48 * 48: n * 2 * 4 * 48 * 48 + n * 2 * 4 * 24 * 24].view(n, 2, 4, 24, 24).clone()
    return a, b, p0.clone(), p1.clone()


def timed(gray, ids, reps=10):
    f = lambda: pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64, ids_checked=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 5)(); work = (ctypes.c_double * 5)(); launches = (ctypes.c_int64 * 5)()
    L.mm_profile_begin()
    for _ in range(reps):
        f()
    L.mm_profile_end(ms, work, launches)
    return ms[1] / reps, ms[2] / reps


names = ["mag", "B", "R", "phase"]
for clips in [int(a) for a in sys.argv[1:]] or [1, 3, 5, 9, 17, 32, 64]:
    n = clips * 64
    base = np.concatenate([synthetic.preprocess_host(synthetic.make_clip_u8(c, 64))[0] for c in range(min(clips, 32))])
    gray = torch.from_numpy(base).to(dev).repeat((clips + 31) // 32, 1, 1)[:n].contiguous()
    if clips == 3:
        gray[5] = 0.0          # degenerate frames: constant, and one with a huge dynamic range
        gray[6] = 7.5
        gray[7] *= 1e4
    ids = torch.from_numpy(np.concatenate([sampler.window_ids(0, 64, 64) + 64 * c for c in range(clips)]).astype(np.int32)).to(dev)
    with torch.no_grad():
        os.environ["MM_PF_WAVE"] = "0"
        a0, b0, p00, p10 = planes(gray, ids)
        t0 = timed(gray, ids)
        os.environ["MM_PF_WAVE"] = "1"
        a1, b1, p01, p11 = planes(gray, ids)
        t1 = timed(gray, ids)
    line = []
    for lvl, (u, v) in enumerate([(a0, a1), (b0, b1)]):
        for k in range(4):
            x, y = u[:, :, k], v[:, :, k]
            same = torch.equal(x.view(torch.int32), y.view(torch.int32))
            d = (x - y).abs()
            rel = (d / x.abs().clamp_min(1e-30)).max().item()
            line.append("L%d %s %s%s" % (lvl + 1, names[k], "bit-equal" if same else "DIFF max abs %.3g rel %.3g (%d px)" % (d.max().item(), rel, int((d > 0).sum())),
                                         "" if torch.isfinite(y).all() else " NONFINITE"))
    out_same = torch.equal(p00.view(torch.int32), p01.view(torch.int32)) and torch.equal(p10.view(torch.int32), p11.view(torch.int32))
    print("%3d clips (%5d frames): frame kernel %.4f -> %.4f ms (x%.2f), window kernels %.4f / %.4f ms | phase outputs %s | %s"
          % (clips, n, t0[0], t1[0], t0[0] / t1[0], t0[1], t1[1], "bit-equal" if out_same else "DIFFER", "; ".join(line)), flush=True)
