#!/usr/bin/env python
"""why does the stem run at a 1.99 GHz effective clock?  The 7x7/2 stem alone, 2 048 frames, in its two input
forms -- packed NHWC3 rows with 4-byte-aligned dwordx4 DMA (KMODE 5, K = 168, what the bench runs) and NHWC4 with 16-byte-aligned
quads (KMODE 4, K = 196) -- so that a PMC pass (tools/pmc_kernel.sh with PK_DRIVER=this file) shows clock, matrix-pipe busy and
stall split of each.  Also prints hipEvent times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import weights  # noqa: E402
from mimamo_net_amd.resnet50_extractor import Resnet50_Extractor  # noqa: E402

dev = torch.device("cuda:0")
net = Resnet50_Extractor(state_dict=weights.make_resnet50_state_dict(seed=0), device=dev)
n = int(sys.argv[1]) * 64 if len(sys.argv) > 1 else 2048
g = torch.Generator(device="cpu").manual_seed(1)
x3 = torch.zeros(n, 230, 230, 3, device=dev)
x3[:, 3:227, 3:227] = (torch.rand(n, 224, 224, 3, generator=g) * 255 - 110).to(dev)
x4 = torch.zeros(n, 224, 224, 4, device=dev)
x4[..., :3] = x3[:, 3:227, 3:227]
for name, x, kw in (("packed NHWC3 (KMODE 5)", x3, {}), ("NHWC4 (KMODE 4)", x4, {"channels_last4": True})):
    with torch.no_grad():
        for _ in range(2):
            net.get_vec(x, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            net.get_vec(x, **kw)
        e1.record()
        torch.cuda.synchronize()
    print("%s: whole trunk %.2f ms per %d frames" % (name, e0.elapsed_time(e1) / 3, n))
