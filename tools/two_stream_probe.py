"""Probe: does running two half-batches on two HIP streams beat one full batch on one stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mimamo_net_amd
from mimamo_net_amd import weights, synthetic
from mimamo_net_amd.pipeline import HotPath

dev = torch.device("cuda:0")
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hsd, rsd = weights.make_two_stream_state_dict(0), weights.make_resnet50_state_dict(0)
lanes = [HotPath(hsd, rsd, dev) for _ in range(NL)]
one = HotPath(hsd, rsd, dev)
frames = torch.from_numpy(np.concatenate([synthetic.make_clip_u8(c, 64) for c in range(4)])).to(dev).repeat(clips // 4, 1, 1, 1, 1).reshape(-1, 112, 112, 3)
full_plan = one.plan([64] * clips)
half_plan = [l.plan([64] * (clips // NL)) for l in lanes]
half = [c.contiguous() for c in frames.chunk(NL)]
streams = [torch.cuda.Stream() for _ in range(NL)]

def step_one():
    return one.forward_u8(frames, full_plan, independent_clips=True)

def step_two():
    outs = []
    for l, p, f, s in zip(lanes, half_plan, half, streams):
        with torch.cuda.stream(s):
            outs.append(l.forward_u8(f, p, independent_clips=True))
    return outs

for name, fn in (("one stream", step_one), ("%d streams" % NL, step_two), ("one stream", step_one), ("%d streams" % NL, step_two)):
    with torch.no_grad():
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.1f ms/step  %.0f frames/s" % (name, dt / 5 * 1e3, clips * 64 * 5 / dt))
a = step_one(); b = torch.cat(step_two()); torch.cuda.synchronize()
print("max diff", (a - b).abs().max().item())
