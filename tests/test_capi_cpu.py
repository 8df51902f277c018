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
    assert L.mm_version() == 106
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


def test_fragment_ordered_masks_are_a_relayout_of_the_plain_ones(L):
    """Round 6: pyramid_wave_kernel (csrc/pyramid_wave.hip) reads the four band masks in MFMA-fragment order -- the complex value lane
    (li, lk) multiplies into its A fragment of k-step ks of 16-row tile tr sits at [(tr * KS + ks) * 64 + lane], row 16 tr + frag_row(li),
    column 4 ks + lk, band 1 read transposed -- and skips the last three k-steps of the first and the last level-1 tile row as exactly
    zero.  Both are properties of the packed host tables (mm_pyramid_host_tables, no GPU): the fragment tables are a pure re-layout of the
    plain tables the round-3 kernels read, and the skipped blocks are zero."""
    n = L.mm_pyramid_host_tables(48, 4, 2, 2, None, 0)
    assert n > 0
    buf = (ctypes.c_float * n)()
    assert L.mm_pyramid_host_tables(48, 4, 2, 2, buf, n) == n
    assert L.mm_pyramid_host_tables(48, 4, 2, 2, buf, n - 1) == -6                      # MM_ERR_WORKSPACE
    assert L.mm_pyramid_host_tables(48, 4, 4, 2, None, 0) == -3                         # unsupported configuration
    t = np.frombuffer(buf, dtype=np.float32)
    S = 48
    off = {"dct": 0, "ec": S * S, "es": 2 * S * S, "m1b0": 3 * S * S}                   # csrc/pyramid_tables.h
    off["m1b1"] = off["m1b0"] + 96 * 48 * 2
    off["m2b0"] = off["m1b1"] + 96 * 48 * 2
    off["m2b1"] = off["m2b0"] + 48 * 24 * 2
    off["f1b0"] = off["m2b1"] + 48 * 24 * 2
    off["f1b1"] = off["f1b0"] + 96 * 48 * 2
    off["f2b0"] = off["f1b1"] + 96 * 48 * 2
    off["f2b1"] = off["f2b0"] + 48 * 24 * 2
    assert n == off["f2b1"] + 48 * 24 * 2
    frag_row = lambda li: 4 * (li & 3) + (li >> 2)
    assert sorted(frag_row(li) for li in range(16)) == list(range(16))
    for level, H in ((1, 48), (2, 24)):
        KS, NTR = H // 4, 2 * H // 16
        for band in (0, 1):
            plain = t[off["m%db%d" % (level, band)]:][:2 * H * H * 2].reshape((2 * H, H, 2) if band == 0 else (H, 2 * H, 2))
            frag = t[off["f%db%d" % (level, band)]:][:2 * H * H * 2].reshape(NTR, KS, 64, 2)
            assert np.abs(plain).max() > 0
            for tr in range(NTR):
                for ks in range(KS):
                    for lane in range(64):
                        row, col = 16 * tr + frag_row(lane & 15), 4 * ks + (lane >> 4)
                        want = plain[row, col] if band == 0 else plain[col, row]
                        assert (frag[tr, ks, lane] == want).all(), (level, band, tr, ks, lane)
            if level == 1:      # EDGE_ZERO_KSTEPS = 3: the spectrum beyond radius 48
                assert np.abs(frag[0, KS - 3:]).max() == 0.0 and np.abs(frag[NTR - 1, KS - 3:]).max() == 0.0
                assert np.abs(frag[0, :KS - 3]).max() > 0.0 and np.abs(frag[1, KS - 3:]).max() > 0.0


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


SCF_FULL_CASES = [("a", 96, 4, 2, 1, 8, 1e-6), ("b", 32, 3, 4, 2, 9, 1e-14), ("c", 32, 3, 3, 1, 10, 1e-14),
                  # round 5: odd grids (50 -> 25; 75 -> 38 -> 19; 84 -> 42 -> 21)
                  ("d", 50, 3, 2, 1, 11, 1e-14), ("e", 75, 4, 2, 1, 12, 1e-14), ("f", 84, 4, 2, 1, 13, 1e-6)]


def _scf_table(L, size, height, nbands, index):
    side, cp = ctypes.c_int(), ctypes.c_int()
    assert L.mm_scfpyr_host_table(size, height, nbands, 2, index, None, ctypes.byref(side), ctypes.byref(cp)) == 0
    t = np.zeros((side.value, side.value, 2))
    assert L.mm_scfpyr_host_table(size, height, nbands, 2, index, t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                  ctypes.byref(side), ctypes.byref(cp)) == 0
    return t[..., 0] + 1j * t[..., 1], bool(cp.value)


@pytest.mark.parametrize("case", SCF_FULL_CASES)
def test_scfpyr_full_host_tables_reproduce_reference_build(L, golden, case):
    """The per-output spectral multipliers of the general pyramid (mm_scfpyr_*), applied with numpy's FFT, give the
    reference's SCFpyr_PyTorch.build outputs (hi residual, every band, lo residual; 2, 3 and 4 bands)."""
    from mimamo_net_amd import weights
    tag, size, height, nbands, n, seed, tol = case
    g = golden("scfpyr_full")
    x = weights.det_uniform("scf." + tag, (n, 1, size, size), 0.0, 1.0, seed)[:, 0].astype(np.float64)
    F = np.fft.fft2(x)
    nout = 2 + (height - 2) * nbands
    for i in range(nout):
        T, is_complex = _scf_table(L, size, height, nbands, i)
        m = T.shape[0]
        k = np.arange(m)
        fa = np.where(k < (m + 1) // 2, k, k - m) % size     # signed frequency (odd m: one more non-negative than negative), modulo the image side
        o = np.fft.ifft2(F[:, fa][:, :, fa] * T) * (m * m)   # the table already carries ifft's 1/m^2
        if i == 0 or i == nout - 1:
            assert not is_complex
            want, got = g[tag + ("_hi" if i == 0 else "_lo")], o.real
        else:
            assert is_complex
            want, got = g["%s_l%d" % (tag, (i - 1) // nbands + 1)][(i - 1) % nbands], np.stack([o.real, o.imag], -1)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max()), (i, np.abs(got - want).max())


def test_scfpyr_config_errors(L):
    side, cp = ctypes.c_int(), ctypes.c_int()
    q = lambda *a: L.mm_scfpyr_host_table(*a, 0, None, ctypes.byref(side), ctypes.byref(cp))
    assert q(96, 5, 2, 2) == -2      # 5 > floor(log2 96) - 2 = 4: 'image too small' (SCFpyr_PyTorch.py:90-91)
    assert q(96, 4, 1, 2) == -3      # nbands = 1 never terminates in the reference (quirk Q7)
    assert q(128, 4, 2, 2) == 0 and q(258, 4, 2, 2) == 0 and q(1026, 4, 2, 2) == -3     # up to 1024 (global scratch above the LDS-resident 96)
    assert q(84, 4, 2, 2) == 0 and q(75, 4, 2, 2) == 0      # odd level grids (84 -> 42 -> 21) and odd images since round 5
    assert q(40, 3, 2, 2) == 0 and side.value == 40
    assert q(96, 4, 2, 2) == 0 and side.value == 96 and cp.value == 0
    h = ctypes.c_void_p()
    assert L.mm_scfpyr_create(ctypes.byref(h), 96, 4, 2, 2) == -5  # no device here: fails loudly


def test_weight_blobs_accept_checkpoint_files_as_stored(pkg):
    """torch tensors, BN `num_batches_tracked` counters and the third-party model's unused `classifier.*` keys do not
    change the blobs handed to mm_resnet50_create / mm_head_create (api/utils/model_utils.py:65-79, api/tester.py:47-49)."""
    import torch
    from mimamo_net_amd import weights
    rs = weights.make_resnet50_state_dict(seed=1)
    disk = {k: torch.from_numpy(np.asarray(v)) for k, v in rs.items()}
    disk["classifier.weight"] = torch.zeros(8, 2048, 1, 1)
    disk["classifier.bias"] = torch.zeros(8)
    disk["conv1_7x7_s2_bn.num_batches_tracked"] = torch.tensor(3)
    np.testing.assert_array_equal(weights.resnet50_blob(disk), weights.resnet50_blob(rs))
    hs = weights.make_two_stream_state_dict(seed=1)
    np.testing.assert_array_equal(weights.two_stream_blob({k: torch.from_numpy(np.asarray(v)) for k, v in hs.items()}),
                                  weights.two_stream_blob(hs))
    one = weights.make_two_stream_state_dict(seed=1, n_out=1)
    wide = weights.widen_classifier(one)
    assert wide["classifier.1.weight"].shape == (2, 256) and (wide["classifier.1.weight"][1] == 0).all()
    assert weights.two_stream_blob(wide).size == weights.two_stream_blob(hs).size


def test_measurement_modes_are_not_in_the_default_library(L):
    """Round-3 verdict: the conv engine's ablation instantiation (tile >= 16, results wrong by construction) and the pyramid kernel's
    MM_PF_ABLATE / MM_PF_LDS_PAD switches exist only in a library built with -DMM_MEASURE.  The default build answers any tile
    outside the documented 0..5 with MM_ERR_INVALID_ARG before touching a pointer, and does not read the two variables."""
    fake = ctypes.c_void_p(0x1000)      # never dereferenced: argument validation comes first
    for tile in (6, 15, 16, 17, 144, 2047, 2048):
        rc = L.mm_conv2d_nhwc(fake, fake, None, None, None, None, fake, 1, 8, 8, 64, 64, 0, 64, 64, 0, 64, 3, 3, 1, 1, 1, tile, 1, None)
        assert rc == -1, (tile, rc)
    blob = open(os.path.join(ROOT, "mimamo-net_amd", "libmimamo_hip.so"), "rb").read()
    assert b"MM_PF_ABLATE" not in blob and b"MM_PF_LDS_PAD" not in blob
    # round-4 verdict: tuning leftovers that changed shipped behaviour from the environment are measurement-build-only too
    for knob in (b"MM_STEM_TILE", b"MM_WF_MAX_CIN", b"MM_WINO_FUSED_SHAPE", b"MM_PF_GRID"):
        assert knob not in blob, knob
    assert b"MM_TAIL_SPLIT" in blob      # A/B knobs whose results are correct stay (sanity check of the string search)
    # and the shipped library's flag stamp says so (build.py relinks when a -DMM_MEASURE build was left behind)
    from mimamo_net_amd import build
    assert "MM_MEASURE" not in (build.library_flags() or "")


def test_head_blob_size_query_agrees_with_create_on_error_codes(L):
    units = (ctypes.c_int * 3)(2048, 256, 256)
    assert L.mm_head_blob_floats_cfg(3, units, 12) == L.mm_head_blob_floats() > 0
    assert L.mm_head_blob_floats_cfg(3, units, 6) > 0
    for ok in (5, 33, 40, 128):      # odd / more than 32 differences are built since round 5 (zero-padded channel groups)
        assert L.mm_head_blob_floats_cfg(3, units, ok) > 0, ok
    for bad in (129, 1000):          # what mm_head_create_cfg answers too (header)
        assert L.mm_head_blob_floats_cfg(3, units, bad) == -3, bad
    assert L.mm_head_blob_floats_cfg(3, units, 0) == -1
    assert L.mm_head_blob_floats_cfg(1, units, 12) == -1
