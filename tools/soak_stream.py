"""Soak of the streamed input path: N bench-sized steps whose uint8 frames come from pinned host memory through the copy stream's
double buffer, alternating between several step contents; every step must give the bits of the same content computed from HBM-resident
frames (a slot re-filled before its readers finished, or read before its upload landed, shows up as a mismatch)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import synthetic, weights
from mimamo_net_amd.pipeline import HotPath
from mimamo_net_amd.stream import FrameStream, pin

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
clips = 16
dev = torch.device("cuda:0")
hot = HotPath(weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0), dev)
contents = [pin(np.concatenate([synthetic.make_clip_u8(100 * k + c, 64) for c in range(clips)])) for k in range(3)]
lengths = [64] * clips
bad = 0
with torch.no_grad():
    refs = [hot.forward_lanes((c.to(dev),), lengths, 3, independent_clips=True, from_u8=True).clone() for c in contents]
    fs = FrameStream(dev, clips * 64)
    order = [(i * 7 + i // 5) % 3 for i in range(N + 1)]
    fs.upload(0, [contents[order[0]]])
    for i in range(N):
        slot = i % 2
        fs.upload(1 - slot, [contents[order[i + 1]]])           # next step's frames while this one computes
        frames = fs.acquire(slot)
        out = hot.forward_lanes((frames,), lengths, (1, 2, 3)[i % 3], independent_clips=True, from_u8=True)
        fs.release(slot)
        if not torch.equal(out, refs[order[i]]):
            bad += 1
            print("step %d: %d values differ" % (i, int((out != refs[order[i]]).sum())), flush=True)
    # and the chunked host path of HotPath.forward_u8 (pinned frames, small chunks)
    hot.upload_chunk_frames = 200
    plan = hot.plan(lengths)
    for k in range(3):
        out = hot.forward_u8(contents[k], plan, True)
        if not torch.equal(out, refs[k]):
            bad += 1
            print("chunked host path, content %d: mismatch" % k, flush=True)
torch.cuda.synchronize()
print("soak_stream: %d steps, %d mismatching" % (N, bad))
sys.exit(1 if bad else 0)
