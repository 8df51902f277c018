"""Soak: N bench-sized steps (32 clips x 64 frames, 3 lanes) must all produce the same bits -- catches rare
load-dependent races (the ring-slot WAR race of round 1 showed up in ~10 % of chip-filling launches of one shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import weights
from mimamo_net_amd.pipeline import HotPath

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
hot = HotPath(weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0), dev)
g = torch.Generator(device="cpu").manual_seed(1)
n = 32 * 64
gray = torch.rand(n, 48, 48, generator=g).to(dev)
rgb = (torch.rand(n, 224, 224, 4, generator=g) * 200 - 100).to(dev)
rgb[..., 3] = 0
lengths = [64] * 32
bad = 0
with torch.no_grad():
    ref = hot.forward_lanes((gray, rgb), lengths, 3, independent_clips=True).clone()
    for i in range(N):
        lanes = (1, 2, 3)[i % 3]
        out = hot.forward_lanes((gray, rgb), lengths, lanes, independent_clips=True)
        if not torch.equal(out, ref):
            bad += 1
            print("step %d (lanes %d): %d values differ, max |d| %.3e" % (i, lanes, int((out != ref).sum()), float((out - ref).abs().max())), flush=True)
torch.cuda.synchronize()
print("soak: %d steps, %d mismatching" % (N, bad))
sys.exit(1 if bad else 0)
