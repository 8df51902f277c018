#!/usr/bin/env python
"""Per-launch table of one single-stream step (hipEvent-bracketed by the library's measurement hook, MM_PROF_DUMP):
kernel category, shape tag, launches, ms, TFLOP/s or GB/s.  usage: layer_table.py [clips] [winograd mode]"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = os.environ.setdefault("MM_PROF_DUMP", "/tmp/mm_prof_dump.csv")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import _lib, synthetic, weights  # noqa: E402
from mimamo_net_amd.pipeline import HotPath  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
hot = HotPath(weights.make_two_stream_state_dict(0), weights.make_resnet50_state_dict(0), dev)
hot.resnet.set_winograd(mode)
if os.environ.get("LT_PRECISION"):          # e.g. LT_PRECISION=bf16x3 (bench.py's extra.bf16x3 schedule)
    hot.resnet.set_precision(os.environ["LT_PRECISION"])
one = synthetic.make_clip_u8(0, 64)
frames = torch.from_numpy(np.concatenate([one] * clips)).to(dev)
plan = hot.plan([64] * clips)
L = _lib.lib()
with torch.no_grad():
    for _ in range(2):
        hot.forward_u8(frames, plan, True)
    torch.cuda.synchronize()
    L.mm_profile_begin()
    hot.forward_u8(frames, plan, True)
    ms, wk, ln = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    assert L.mm_profile_end(ms, wk, ln) == 0
agg = collections.OrderedDict()
for line in open(dump):
    cat, work, t, tag = line.rstrip("\n").split(",", 3)
    a = agg.setdefault((int(cat), tag), [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(work); a[2] += float(t)
names = {0: "conv", 1: "pyramid", 2: "window", 3: "wino-xf", 4: "other"}
tot = collections.Counter()
for (cat, tag), (n, work, t) in agg.items():
    rate = work / (t * 1e-3) / 1e12 if cat == 0 else work / (t * 1e-3) / 1e9
    print("%-8s %-48s x%-3d %8.3f ms  %8.1f %s" % (names[cat], tag, n, t, rate, "TFLOP/s" if cat == 0 else "GB/s"))
    tot[cat] += t
print("totals (ms):", {names[c]: round(v, 2) for c, v in tot.items()}, "frames", clips * 64)
