"""Import the (read-only) Python reference at /root/reference/api with in-memory shims.

TEST INFRASTRUCTURE ONLY.  This module exists so that `make_golden.py` can run the
*real* reference code in the build container and freeze its outputs as small data
fixtures under tests/golden/.  It is never imported by the product package, by the
`-m gpu` tests, by `bench.py` or by `smoke()` -- /root/reference does not exist on
the GPU box.  Nothing from the reference is copied: the reference modules are
loaded from where they lie, with five shims for APIs that have since been removed
from numpy / torch:

  1. np.complex                      (used SCFpyr_PyTorch.py:64-65)
  2. torch.rfft / torch.ifft         (old signal_ndim API, SCFpyr_PyTorch.py:110,122,133,171)
  3. a stub `torchvision` module     (steerable/utils.py:25 imports it; only get_device is used)
  4. utils/phase_utils.py:107 uses the py<3.7 keyword `async=` inside a dead function;
     the token is rewritten in memory before exec
  5. matplotlib Agg backend
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF_API = "/root/reference/api"


def available():
    return os.path.isdir(REF_API)


def _install_shims():
    sys.dont_write_bytecode = True
    if not hasattr(np, "complex"):
        np.complex = complex  # shim 1

    if not hasattr(torch, "rfft"):  # shim 2
        def rfft(x, signal_ndim=2, onesided=False):
            assert signal_ndim == 2 and not onesided
            return torch.view_as_real(torch.fft.fft2(x))

        def ifft(x, signal_ndim=2):
            assert signal_ndim == 2
            return torch.view_as_real(torch.fft.ifft2(torch.view_as_complex(x.contiguous())))

        torch.rfft = rfft
        torch.ifft = ifft

    if "torchvision" not in sys.modules:  # shim 3
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")

        class _Stub(object):
            def __init__(self, *a, **k):
                pass

        class Resize(object):
            """Stand-in for the absent torchvision.transforms.Resize on PIL images:
            int size -> smaller edge scaled to `size`, aspect kept; PIL does the resampling."""
            def __init__(self, size, interpolation=2):
                self.size, self.interpolation = size, interpolation

            def __call__(self, img):
                w, h = img.size
                if isinstance(self.size, int):
                    if w <= h:
                        ow, oh = self.size, int(self.size * h / w)
                    else:
                        oh, ow = self.size, int(self.size * w / h)
                else:
                    oh, ow = self.size
                return img.resize((ow, oh), self.interpolation)

        class Compose(object):
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, x):
                for t in self.ts:
                    x = t(x)
                return x

        for name in ("ToPILImage", "Grayscale", "RandomCrop", "CenterCrop", "ToTensor", "Normalize"):
            setattr(tvt, name, _Stub)
        tvt.Resize = Resize
        tvt.Compose = Compose
        tv.transforms = tvt
        tv.utils = types.ModuleType("torchvision.utils")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
        sys.modules["torchvision.utils"] = tv.utils

    import matplotlib
    matplotlib.use("Agg")  # shim 5


def _load_phase_utils():
    """shim 4: exec utils/phase_utils.py with `async=` -> `non_blocking=`."""
    path = os.path.join(REF_API, "utils", "phase_utils.py")
    src = open(path).read().replace("async=True", "non_blocking=True")
    pkg = types.ModuleType("utils")
    pkg.__path__ = [os.path.join(REF_API, "utils")]
    sys.modules.setdefault("utils", pkg)
    mod = types.ModuleType("utils.phase_utils")
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)
    sys.modules["utils.phase_utils"] = mod
    return mod


class Ref(object):
    """Handles onto the reference's hot-path classes/functions."""
    pass


def load():
    if not available():
        raise RuntimeError("reference not present at %s" % REF_API)
    _install_shims()
    if REF_API not in sys.path:
        sys.path.insert(0, REF_API)
    saved_dtype = torch.get_default_dtype()
    ref = Ref()
    ref.phase_utils = _load_phase_utils()
    import steerable.SCFpyr_PyTorch as scf_t
    import steerable.SCFpyr_NumPy as scf_n
    import steerable.math_utils as mu
    import phase_difference_extractor as pde
    import mimamo_net as net
    ref.SCFpyr_PyTorch = scf_t.SCFpyr_PyTorch
    ref.scf_torch_module = scf_t
    ref.SCFpyr_NumPy = scf_n.SCFpyr_NumPy
    ref.math_utils = mu
    ref.Phase_Difference_Extractor = pde.Phase_Difference_Extractor
    ref.Two_Stream_RNN = net.Two_Stream_RNN
    ref.mimamo_net = net
    import sampler.snippet_sampler as ss
    ref.Snippet_Sampler = ss.Snippet_Sampler
    import tester as tester_mod
    ref.Tester = tester_mod.Tester
    torch.set_default_dtype(saved_dtype)
    return ref
