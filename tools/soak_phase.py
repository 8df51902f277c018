"""Soak of the phase stage alone (round 6: pyramid_wave_kernel + padded-plane window kernels): N calls of phase_diff_frames over a cycle of batch
sizes (whole rounds, remainders on either pyramid kernel, every workgroup shape), alternating between two streams that run concurrently, must
reproduce the bits of the first call of each size -- a load-dependent race in the wave-private LDS hand-overs would show here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd.phase_difference_extractor import Phase_Difference_Extractor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
pde = Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
g = torch.Generator(device="cpu").manual_seed(3)
sizes = [2048, 683, 2112, 64, 1500, 4096, 300, 1024, 2048 + 600]
frames = {n: torch.rand(n, 48, 48, generator=g).to(dev) for n in sizes}
ids = {n: torch.clamp(torch.arange(n, device=dev)[:, None] + torch.arange(-6, 7, device=dev)[None, :], 0, n - 1).int().contiguous() for n in sizes}
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ref = {}
bad = 0
with torch.no_grad():
    for n in sizes:
        a, b = pde.phase_diff_frames(frames[n], ids[n], nhwc=True, ids_checked=True)
        ref[n] = (a.clone(), b.clone())
    torch.cuda.synchronize()
    pending = []
    for i in range(N):
        n = sizes[i % len(sizes)]
        with torch.cuda.stream(streams[i % 2]):
            a, b = pde.phase_diff_frames(frames[n], ids[n], nhwc=True, ids_checked=True)
            pending.append((i, n, a, b))
        if len(pending) >= 8:
            torch.cuda.synchronize()
            for k, m, x, y in pending:
                if not (torch.equal(x, ref[m][0]) and torch.equal(y, ref[m][1])):
                    bad += 1
                    print("call %d (n = %d): differs, max |d| %.3e / %.3e" % (k, m, float((x - ref[m][0]).abs().max()), float((y - ref[m][1]).abs().max())), flush=True)
            pending = []
torch.cuda.synchronize()
print("soak_phase: %d calls over sizes %s on two streams, %d mismatching" % (N, sizes, bad))
sys.exit(1 if bad else 0)
