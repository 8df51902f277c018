"""Experiment: do free-running, skewed lanes (no per-step join) overlap the HBM-bound early ResNet layers of one lane
with the MFMA-bound late layers of the other?  Compares, on the same box:
  joined   : forward_lanes as bench.py uses it (lanes fork/join every step)
  free     : each lane loops over its own clips on its own stream, no cross-lane waits
  free+skew: as free, but lane 1 starts half a ResNet pass late"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import weights
from mimamo_net_amd.pipeline import HotPath

dev = torch.device("cuda:0")
CLIPS, STEPS, LANES = int(os.environ.get("CLIPS", 32)), int(os.environ.get("STEPS", 8)), int(os.environ.get("LANES", 2))
hot = HotPath(weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0), dev)
N = CLIPS * 64
gray = torch.rand(N, 48, 48, device=dev)
rgb = torch.rand(N, 224, 224, 4, device=dev) * 100 - 50
lengths = [64] * CLIPS


def joined(steps):
    for _ in range(steps):
        out = hot.forward_lanes((gray, rgb), lengths, LANES, independent_clips=True)
    return out


per = CLIPS // LANES
plans = [hot.plan([64] * per) for _ in range(LANES)]
streams = [torch.cuda.Stream(device=dev) for _ in range(LANES)]


def free(steps, skew):
    cur = torch.cuda.current_stream()
    for l, st in enumerate(streams):
        st.wait_stream(cur)
    if skew:
        for l in range(1, LANES):
            with torch.cuda.stream(streams[l]):
                hot.resnet.get_vec(rgb[: int(per * 64 * skew * l / LANES * 2)], channels_last4=True)
    outs = []
    for _ in range(steps):
        for l, st in enumerate(streams):
            with torch.cuda.stream(st):
                f0, f1 = l * per * 64, (l + 1) * per * 64
                outs.append(hot.forward(gray[f0:f1], rgb[f0:f1], plans[l], independent_clips=True))
    for st in streams:
        cur.wait_stream(st)
    return outs


def timeit(fn, *a):
    fn(*a)  # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(*a)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


with torch.no_grad():
    for name, fn, a in (("joined", joined, (STEPS,)), ("free", free, (STEPS, 0.0)), ("free+skew0.5", free, (STEPS, 0.5)),
                        ("joined", joined, (STEPS,)), ("free+skew0.25", free, (STEPS, 0.25)), ("free", free, (STEPS, 0.0))):
        dt = timeit(fn, *a)
        print("%-14s %7.2f ms/step  %8.0f frames/s" % (name, dt / STEPS * 1e3, N * STEPS / dt), flush=True)
