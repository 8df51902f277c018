"""Per-stage throughput for BASELINE configs[1] (steerable pyramid + phase difference only, 64-frame clips) and
configs[2] (ResNet50 pool5 extractor on 224x224 batches) on one MI355X; configs[3] is bench.py itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import weights, sampler
from mimamo_net_amd.pipeline import HotPath

dev = torch.device("cuda:0")
hot = HotPath(weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0), dev)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


with torch.no_grad():
    print("configs[1]  pyramid + phase difference only (gray 48x48 frames resident in HBM, 13-frame windows, NHWC outputs)")
    for clips in (1, 32, 256):
        n = clips * 64
        gray = torch.rand(n, 48, 48, device=dev)
        # window ids of ALL clips (plan() splits the rows into head-call groups of 64 snippets: up to round 2 this tool handed
        # over groups[0] only, i.e. at 256 clips the window kernels ran for the first 64 clips -- its 3.17 M frames/s was inflated)
        import numpy as np
        ids = torch.from_numpy(np.concatenate([sampler.window_ids(0, 64, 64) + 64 * c for c in range(clips)]).astype(np.int32)).to(dev)
        dt = timeit(lambda: hot.pde.phase_diff_frames(gray, ids, nhwc=True, out1_cstride=88, out1_coffset=64, ids_checked=True), 20)
        print("   %4d clips (%6d frames): %.3f ms  %.2f M frames/s  %.0f GB/s algorithmic" % (clips, n, dt * 1e3, n / dt / 1e6, n * 285696 / dt / 1e9))
    print("configs[2]  ResNet50 pool5 extractor (fp32 NCHW batches resident in HBM)")
    for bs in (64, 256, 1024):
        x = torch.rand(bs, 3, 224, 224, device=dev) * 255 - 110
        dt = timeit(lambda: hot.resnet.get_vec(x), 10)
        print("   batch %4d, one stream : %.2f ms  %.0f frames/s" % (bs, dt * 1e3, bs / dt))
    x = torch.rand(2048, 3, 224, 224, device=dev) * 255 - 110
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    def lanes3():
        cur = torch.cuda.current_stream()
        for i, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                hot.resnet.get_vec(x[i * 683:(i + 1) * 683] if i < 2 else x[1366:])
        for st in streams:
            cur.wait_stream(st)
    dt = timeit(lanes3, 6)
    print("   batch 2048, three streams: %.2f ms  %.0f frames/s" % (dt * 1e3, 2048 / dt))
