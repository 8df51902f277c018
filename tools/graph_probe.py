"""Feasibility/latency probe: capture one pass of the hot path (all launches of libmimamo_hip.so on the capturing
stream) in a HIP graph through torch.cuda.CUDAGraph and replay it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import weights
from mimamo_net_amd.pipeline import HotPath

dev = torch.device("cuda:0")
hot = HotPath(weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0), dev)
for clips in (1, 2, 4):
    n = clips * 64
    gray = torch.rand(n, 48, 48, device=dev)
    rgb = torch.rand(n, 224, 224, 4, device=dev) * 100 - 50
    plan = hot.plan([64] * clips)
    with torch.no_grad():
        for _ in range(3):
            ref = hot.forward(gray, rgb, plan, independent_clips=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ref = hot.forward(gray, rgb, plan, independent_clips=True)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            hot.forward(gray, rgb, plan, independent_clips=True)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = hot.forward(gray, rgb, plan, independent_clips=True)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, ref), (out - ref).abs().max()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 20
    print("clips %d: eager %.3f ms (%.0f frames/s)   graph replay %.3f ms (%.0f frames/s)" % (clips, eager * 1e3, n / eager, graph * 1e3, n / graph), flush=True)
