"""Random-shape fuzz of the conv/GEMM engine (mm_conv2d_nhwc) against float64 torch on the host: ragged tiles, K tails,
strides, paddings, channel windows, residual / ReLU / post-affine, both K orders, every tile shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
torch.set_num_threads(32)


def pack(w, korder):
    co, ci, kh, kw = w.shape
    k = kh * kw * ci
    kp = (k + 15) // 16 * 16
    out = np.zeros((co, kp), dtype=np.float32)
    if korder == 0:
        out[:, :k] = w.transpose(0, 2, 3, 1).reshape(co, k)
    else:
        out[:, :k] = w.reshape(co, ci // 16, 16, kh, kw).transpose(0, 1, 3, 4, 2).reshape(co, k)
    return out


worst = 0.0
for it in range(N):
    k = int(rng.choice([1, 1, 3, 3, 5, 7]))
    st = int(rng.choice([1, 1, 2]))
    pad = int(rng.choice([0, k // 2])) if k > 1 else 0
    Ci = 4 * int(rng.randint(1, 40))
    Co = int(rng.choice([4 * int(rng.randint(1, 70)), int(rng.randint(1, 9))]))   # sometimes a narrow (non-multiple-of-4) output
    H = int(rng.randint(max(1, k - 2 * pad), 30)); W = int(rng.randint(max(1, k - 2 * pad), 30))
    B = int(rng.randint(1, 6))
    if (H + 2 * pad - k) < 0 or (W + 2 * pad - k) < 0:
        continue
    korder = int(rng.randint(0, 2)) if (Ci % 16 == 0 and k > 1) else 0
    tile = int(rng.choice([0, 1, 2, 3, 4, 5] if (k == 1 and pad == 0) else [0, 1, 2, 3, 4]))
    relu = int(rng.randint(0, 2)); use_res = int(rng.randint(0, 2)); use_post = int(rng.randint(0, 2)); use_bias = int(rng.randint(0, 2))
    wide = Co % 4 == 0
    in_off = 4 * int(rng.randint(0, 3)); in_cs = Ci + in_off + 4 * int(rng.randint(0, 3))
    out_off = (4 * int(rng.randint(0, 3))) if wide else int(rng.randint(0, 3)); out_cs = Co + out_off + ((4 * int(rng.randint(0, 3))) if wide else int(rng.randint(0, 3)))
    x = (rng.rand(B, Ci, H, W).astype(np.float32) - 0.5)
    w = (rng.rand(Co, Ci, k, k).astype(np.float32) - 0.5) / np.sqrt(Ci * k * k)
    b = rng.rand(Co).astype(np.float32) - 0.5
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double() if use_bias else None, stride=st, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = rng.rand(B, Co, Ho, Wo).astype(np.float32) - 0.5
    ps = rng.rand(Co).astype(np.float32) + 0.5; pt = rng.rand(Co).astype(np.float32) - 0.5
    if use_res: ref = ref + torch.from_numpy(res).double()
    if relu: ref = F.relu(ref)
    if use_post: ref = ref * torch.from_numpy(ps).double()[None, :, None, None] + torch.from_numpy(pt).double()[None, :, None, None]
    ref = ref.numpy()
    xin = torch.zeros(B, H, W, in_cs); xin[..., in_off:in_off + Ci] = torch.from_numpy(x).permute(0, 2, 3, 1); xin = xin.to(dev)
    out = torch.full((B, Ho, Wo, out_cs), -7.0, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    wd = t(pack(w, korder)); bd = t(b) if use_bias else None; rd = t(res.transpose(0, 2, 3, 1)) if use_res else None
    psd, ptd = (t(ps), t(pt)) if use_post else (None, None)
    rc = L.mm_conv2d_nhwc(_lib.ptr(xin), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(psd), _lib.ptr(ptd), _lib.ptr(out),
                          B, H, W, Ci, in_cs, in_off, Co, out_cs, out_off, Co, k, k, st, pad, relu, tile, korder, _lib.current_stream())
    desc = dict(B=B, H=H, W=W, Ci=Ci, Co=Co, k=k, st=st, pad=pad, korder=korder, tile=tile, relu=relu, res=use_res, post=use_post, bias=use_bias,
                in_cs=in_cs, in_off=in_off, out_cs=out_cs, out_off=out_off)
    assert rc == 0, (rc, desc)
    got = out.cpu().numpy()
    err = float(np.abs(got[..., out_off:out_off + Co].transpose(0, 3, 1, 2) - ref).max())
    untouched = np.delete(got, np.s_[out_off:out_off + Co], axis=-1)
    assert (untouched == -7.0).all(), ("wrote outside the channel window", desc)
    assert err < 3e-5, (err, desc)
    worst = max(worst, err)
print("conv fuzz: %d shapes ok, worst abs err %.2e" % (N, worst))
