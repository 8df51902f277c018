"""CPU-side checks of the native library: it loads, exports every symbol include/mimamo_hip.h declares,
its host-side (float64) mask builder reproduces the reference's masks, and it fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L(pkg):
    from mimamo_net_amd import build, _lib
    build.build_library()
    return _lib.lib()


def test_exports_match_header(L):
    from mimamo_net_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mimamo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mm_version() == 100
    assert b"too small" in L.mm_status_string(-2)


def _mask(L, level, band, side):
    out = (ctypes.c_double * (side * side))()
    crop = (ctypes.c_int * 2)()
    assert L.mm_pyramid_host_mask(48, 4, 2, level, band, out, crop) == 0
    return np.array(out).reshape(side, side), list(crop)


def test_host_masks_match_reference(L, golden):
    g = golden("masks")
    for b in range(2):
        m, crop = _mask(L, 1, b, 96)
        ref = g["lo0"] * g["himask_0"] * g["anglemask_0_%d" % b]
        np.testing.assert_allclose(m, ref, rtol=0, atol=2e-15)
        assert crop == [0, 96]
        m, crop = _mask(L, 2, b, 48)
        s, e = int(g["crop_0"][0]), int(g["crop_0"][1])
        ref = g["lo0"][s:e, s:e] * g["lomask_0"] * g["himask_1"] * g["anglemask_1_%d" % b]
        np.testing.assert_allclose(m, ref, rtol=0, atol=2e-15)
        assert crop == [s, e] == [24, 72]  # integer crop bounds: bit-exact


def test_half_plane_support_is_exact(golden):
    """The kernels keep only fv>=0 (band 0) / fu>=0 (band 1): the reference masks are exactly 0 elsewhere."""
    g = golden("masks")
    assert np.abs(g["anglemask_0_0"][:, :48]).max() == 0.0 and np.abs(g["anglemask_0_1"][:48, :]).max() == 0.0
    assert np.abs(g["anglemask_1_0"][:, :24]).max() == 0.0 and np.abs(g["anglemask_1_1"][:24, :]).max() == 0.0
    # Nyquist row/column (|f| = 48) is cut by lo0
    assert np.abs(g["lo0"][0]).max() == 0.0 and np.abs(g["lo0"][:, 0]).max() == 0.0


def test_config_validation_and_no_silent_fallback(L):
    import torch
    h = ctypes.c_void_p()
    assert L.mm_pyramid_create(ctypes.byref(h), 48, 5, 2, 2) == -2   # too small (SCFpyr_PyTorch.py:90-91)
    assert L.mm_pyramid_create(ctypes.byref(h), 48, 4, 4, 2) == -3   # unsupported
    assert L.mm_pyramid_create(ctypes.byref(h), 0, 4, 2, 2) == -1
    if not torch.cuda.is_available():
        assert L.mm_pyramid_create(ctypes.byref(h), 48, 4, 2, 2) == -5  # no device: error, not a CPU path
        from mimamo_net_amd.phase_difference_extractor import Phase_Difference_Extractor
        with pytest.raises(RuntimeError):
            Phase_Difference_Extractor(4, 2, 2, [1, 2]).build_pyramid(torch.zeros(1, 13, 48, 48))
