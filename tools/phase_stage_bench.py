"""configs[1] only (pyramid + phase difference on 64-frame clips): per-kernel times of the fused phase stage from the library's
hipEvent hook, on realistic textured clips (wrap rate matters: the window kernel skips blurs until a window's first wrap)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib, synthetic, sampler
from mimamo_net_amd.phase_difference_extractor import Phase_Difference_Extractor

dev = torch.device("cuda:0")
pde = Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
L = _lib.lib()
for clips in [int(a) for a in sys.argv[1:]] or [1, 11, 32, 256]:
    n = clips * 64
    base = np.concatenate([synthetic.preprocess_host(synthetic.make_clip_u8(c, 64))[0] for c in range(min(clips, 32))])
    gray = torch.from_numpy(base).to(dev).repeat((clips + 31) // 32, 1, 1)[:n].contiguous()
    ids = torch.from_numpy(np.concatenate([sampler.window_ids(0, 64, 64) + 64 * c for c in range(clips)]).astype(np.int32)).to(dev)
    f = lambda: pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64, ids_checked=True)
    with torch.no_grad():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        ms = (ctypes.c_double * 5)(); work = (ctypes.c_double * 5)(); launches = (ctypes.c_int64 * 5)()
        L.mm_profile_begin()
        for _ in range(5):
            f()
        L.mm_profile_end(ms, work, launches)
    print("%4d clips (%6d frames): wall %.3f ms  %.2f M frames/s  %.0f GB/s algorithmic = %.3f of 8 TB/s | per-frame kernel %.3f ms, "
          "window kernels %.3f ms" % (clips, n, dt * 1e3, n / dt / 1e6, n * 285696 / dt / 1e9, n * 285696 / dt / 8e12, ms[1] / 5, ms[2] / 5))
