"""Fuzz of the general steerable pyramid (SCFpyr_PyTorch.build drop-in) and the generic extract kernel against the oracle
over random sizes / heights / band counts / window lengths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mimamo_net_amd  # noqa: F401
from mimamo_net_amd import _lib
from mimamo_net_amd.scfpyr import SCFpyr_PyTorch
from mimamo_net_amd.phase_difference_extractor import Phase_Difference_Extractor
import mm_oracle as o

dev = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
done = 0
for it in range(60):
    size = 2 * int(rng.randint(8, 81))
    hmax = int(np.floor(np.log2(size))) - 2
    if hmax < 2:
        continue
    height = int(rng.randint(2, hmax + 1)); nbands = int(rng.randint(2, 7))
    x = rng.rand(2, 1, size, size)
    pyr = SCFpyr_PyTorch(height, nbands, 2, device=dev, precision=64)
    try:
        coeff = pyr.build(torch.from_numpy(x).to(dev))
    except NotImplementedError:
        # an odd level grid: the oracle's torch.fft path handles it, the HIP path declares it unsupported
        continue
    levels, hi, lo = o.pyramid_build(x[:, 0], height, nbands, dtype=np.float64, keep_residuals=True)
    errs = [np.abs(coeff[0].cpu().numpy() - hi).max(), np.abs(coeff[-1].cpu().numpy() - lo).max()]
    for l, c in enumerate(levels):
        for b in range(nbands):
            errs.append(np.abs(coeff[l + 1][b].cpu().numpy() - np.stack([c[b].real, c[b].imag], -1)).max())
    assert max(errs) < 1e-11, (size, height, nbands, max(errs))
    done += 1
print("scfpyr fuzz: %d configurations ok" % done)
done = 0
for it in range(25):
    W = 4 * int(rng.randint(2, 13)); P = int(rng.randint(2, 17)); nb = int(rng.randint(2, 5)); B = int(rng.randint(1, 3))
    c = (rng.rand(B, nb, P, W, W, 2).astype(np.float32) - 0.5)
    got = Phase_Difference_Extractor(4, nb, 2, 1).extract(torch.from_numpy(c).to(dev)).cpu().numpy()
    want = o.extract(c)
    d = np.abs(got - want)
    # random coefficients have wild phases: allow isolated 2 pi branch decisions to differ, bound everything else
    flips = int((d > 1.0).sum())
    assert got.shape == want.shape and flips <= max(4, d.size // 20000) and np.quantile(d, 0.999) < 1e-3, (W, P, nb, flips, float(np.quantile(d, 0.999)))
    done += 1
print("generic extract fuzz: %d configurations ok" % done)
