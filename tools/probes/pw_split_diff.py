"""Where do the time-split window kernel (MM_PW_SPLIT=2) and the shipped one differ?  (debugging aid for the bit-identity test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import mimamo_net_amd  # noqa
from mimamo_net_amd import synthetic
from mimamo_net_amd.phase_difference_extractor import Phase_Difference_Extractor
dev = torch.device("cuda:0")
pde = Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
base = np.concatenate([synthetic.textured_gray(64, 48, seed=400 + c) for c in range(3)])
n = 192
fast = np.stack([np.roll(base[3 * (t // 64)], (t % 64), axis=1) for t in range(n)]).astype(np.float32)
one = torch.clamp(torch.arange(64, device=dev)[:, None] + torch.arange(-6, 7, device=dev)[None, :], 0, 63)
ids = (one[None] + 64 * torch.arange(3, device=dev)[:, None, None]).reshape(n, 13).int().contiguous()
for name, fr in (("slow", torch.from_numpy(base).to(dev)), ("fast", torch.from_numpy(np.ascontiguousarray(fast)).to(dev))):
    for nhwc in (False, True):
        os.environ.pop("MM_PW_SPLIT", None)
        a = pde.phase_diff_frames(fr, ids, nhwc=nhwc) if not nhwc else pde.phase_diff_frames(fr, ids, nhwc=True, out1_cstride=88, out1_coffset=64)
        a = [t.clone() for t in a]
        os.environ["MM_PW_SPLIT"] = "2"
        b = pde.phase_diff_frames(fr, ids, nhwc=nhwc) if not nhwc else pde.phase_diff_frames(fr, ids, nhwc=True, out1_cstride=88, out1_coffset=64)
        for lvl, (x, y) in enumerate(zip(a, b)):
            if nhwc:
                x, y = x[..., -24:] if lvl else x, y[..., -24:] if lvl else y
                x, y = x.permute(0, 3, 1, 2), y.permute(0, 3, 1, 2)
            ne = (x != y)
            print(name, "nhwc" if nhwc else "nchw", "level", lvl + 1, "differing %.5f max |diff| %.3e" % (ne.float().mean().item(), (x - y).abs().max().item()),
                  "per channel:", [int(v) for v in ne.sum(dim=(0, 2, 3)).tolist()], "windows differing:", int(ne.any(dim=3).any(dim=2).any(dim=1).sum()))
