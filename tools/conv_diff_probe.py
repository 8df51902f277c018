"""Where do two runs of the same launch differ?  (debug aid for the determinism test)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
B, Ci, Co, tile = 400000, 64, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 3
x = torch.rand(B, 1, 1, Ci, device=dev) - 0.5
w = torch.rand(Co, Ci, device=dev) - 0.5
ref64 = (x.view(B, Ci).double() @ w.double().t())
def run():
    out = torch.full((B, 1, 1, Co), float("nan"), device=dev)
    rc = L.mm_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), None, None, None, None, _lib.ptr(out), B, 1, 1, Ci, Ci, 0, Co, Co, 0, Co, 1, 1, 1, 0, 0, tile, 0, _lib.current_stream())
    assert rc == 0
    return out.view(B, Co)
for r in range(12):
    o = run()
    bad = (o.double() - ref64).abs() > 1e-3
    bad |= torch.isnan(o)
    n = int(bad.sum())
    if n:
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("run %d: %d bad words, rows %d..%d (%d rows; row%%64 in %s), cols %s, nan %d" % (r, n, rows.min(), rows.max(), len(rows), sorted(set((rows % 64).tolist()))[:40], cols.tolist()[:70], int(torch.isnan(o).sum())))
        r0 = int(rows[0]); print("   sample row", r0, o[r0, cols[:6]].tolist(), ref64[r0, cols[:6]].tolist())
    else:
        print("run %d ok" % r)
