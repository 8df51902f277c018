"""Shape fuzz of the whole path: videos of awkward lengths (1, 2, 12, 13, 63, 64, 65, 127, 128, 129, 309 ... frames) run
alone and in mixed batches must agree (batch composition only changes GEMM tile choices -> fp32 rounding), stay finite,
and device memory must not creep across repeated calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import weights
from mimamo_net_amd.pipeline import HotPath

dev = torch.device("cuda:0")
hot = HotPath(weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0), dev)
rng = np.random.RandomState(0)
LENGTHS = [1, 2, 3, 12, 13, 14, 63, 64, 65, 100, 127, 128, 129, 200, 309]
g = torch.Generator(device="cpu").manual_seed(3)
clips = {n: (torch.rand(n, 48, 48, generator=g).to(dev), (torch.rand(n, 3, 224, 224, generator=g) * 200 - 100).to(dev)) for n in LENGTHS}
alone = {}
with torch.no_grad():
    for n in LENGTHS:
        gray, rgb = clips[n]
        out = hot.forward(gray, rgb, hot.plan([n]))
        res = hot.assemble(out, hot.plan([n]))[0]
        assert res.shape == (n, 2) and np.isfinite(res).all(), n
        alone[n] = res
    print("alone ok:", LENGTHS, flush=True)
    worst = 0.0
    base_mem = None
    for it in range(40):
        k = rng.randint(2, 6)
        pick = [LENGTHS[i] for i in rng.randint(0, len(LENGTHS), size=k)]
        gray = torch.cat([clips[n][0] for n in pick]); rgb = torch.cat([clips[n][1] for n in pick])
        lanes = int(rng.randint(1, 4))
        if lanes == 1:
            plan = hot.plan(pick)
            res = hot.assemble(hot.forward(gray, rgb, plan), plan)
        else:
            out = hot.forward_lanes((gray, rgb), pick, lanes)
            res = hot.assemble(out, hot.plan(pick))
        for i, n in enumerate(pick):
            err = float(np.abs(res[i] - alone[n]).max())
            worst = max(worst, err)
            assert res[i].shape == (n, 2) and err < 1e-5, (it, pick, i, n, err)
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated()
        if it == 20:
            base_mem = mem
        if base_mem is not None:
            assert mem < base_mem + (8 << 30), ("memory creep", base_mem, mem)
    print("fuzz ok: 40 mixed batches, worst |alone - batched| = %.2e, allocated %.1f GiB" % (worst, torch.cuda.memory_allocated() / 2**30))
