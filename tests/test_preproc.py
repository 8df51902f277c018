"""Frame preprocessing: the library's fixed-point resampling tables + integer passes must be
BIT-EXACT with PIL (the reference's transforms are PIL calls).  CPU part: host tables + a numpy emulation of the
two passes vs PIL itself.  GPU part: the HIP kernels vs PIL on synthetic clips."""
import ctypes

import numpy as np
import pytest

from mimamo_net_amd import synthetic, weights


def _tables(L, n_in, n_out, filt):
    ks = ctypes.c_int()
    assert L.mm_preproc_host_coeffs(n_in, n_out, filt, ctypes.byref(ks), None, None, 0) == 0
    bounds = (ctypes.c_int * (n_out * 2))()
    kk = (ctypes.c_int * (n_out * ks.value))()
    assert L.mm_preproc_host_coeffs(n_in, n_out, filt, ctypes.byref(ks), bounds, kk, n_out * ks.value) == 0
    return np.array(bounds).reshape(n_out, 2), np.array(kk).reshape(n_out, ks.value)


def _resample_axis(img, bounds, kk, axis):
    """One PIL pass: out = clip8((2^21 + sum u8*k) >> 22) along `axis` (uint8 in, uint8 out)."""
    img = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.int64)
    for o, (lo, cnt) in enumerate(bounds):
        acc = (1 << 21) + np.tensordot(kk[o, :cnt].astype(np.int64), img[lo:lo + cnt], axes=(0, 0))
        out[o] = np.clip(acc >> 22, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


@pytest.fixture(scope="module")
def L(pkg):
    from mimamo_net_amd import build, _lib
    build.build_library()
    return _lib.lib()


def test_host_tables_emulation_is_bit_exact_with_pil(L):
    from PIL import Image
    clip = synthetic.make_clip_u8(9, 3)
    noise = (weights.det_uniform("pp.noise", (2, 112, 112, 3), 0, 256, 3)).astype(np.uint8)  # full-range, hits clipping
    frames = np.concatenate([clip, noise])
    lb, lk = _tables(L, 112, 48, 1)
    bb, bk = _tables(L, 112, 256, 0)
    assert lk.shape[1] == 15 and bk.shape[1] == 3
    for f in frames:
        im = Image.fromarray(f, "RGB")
        g = np.asarray(im.convert("L"))
        np.testing.assert_array_equal(synthetic.to_gray_u8(f), g)
        want = np.asarray(im.convert("L").resize((48, 48), Image.LANCZOS))
        got = _resample_axis(_resample_axis(g, lb, lk, 1), lb, lk, 0)
        np.testing.assert_array_equal(got, want)
        want = np.asarray(im.resize((256, 256), Image.BILINEAR))
        got = _resample_axis(_resample_axis(f, bb, bk, 1), bb, bk, 0)
        np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_gpu_preprocessing_bit_exact_with_pil(pkg):
    import torch
    from mimamo_net_amd.preprocess import FramePreprocessor
    dev = torch.device("cuda:0")
    clip = synthetic.make_clip_u8(4, 6)
    noise = (weights.det_uniform("pp.noise", (3, 112, 112, 3), 0, 256, 8)).astype(np.uint8)
    frames = np.concatenate([clip, noise])
    gray_ref, rgb_ref = synthetic.preprocess_host(frames)      # PIL, reference-style
    pp = FramePreprocessor(device=dev)
    g, r4 = pp(torch.from_numpy(frames).to(dev))
    np.testing.assert_array_equal(g.cpu().numpy(), gray_ref)
    r4 = r4.cpu().numpy()
    np.testing.assert_array_equal(r4[..., :3].transpose(0, 3, 1, 2), rgb_ref)
    assert (r4[..., 3] == 0).all()
    _, r = pp(torch.from_numpy(frames).to(dev), channels_last4=False, want_gray=False)
    np.testing.assert_array_equal(r.cpu().numpy(), rgb_ref)
    # zero-bordered packed three-channel layout (the stem's padding in memory): same values inside, exact zeros around
    stale = torch.full((frames.shape[0], 230, 230, 3), 7.0, device=dev)     # the kernel must write the border itself
    del stale
    _, r3 = pp(torch.from_numpy(frames).to(dev), bordered3=True, want_gray=False)
    r3 = r3.cpu().numpy()
    assert r3.shape == (frames.shape[0], 230, 230, 3)
    np.testing.assert_array_equal(r3[:, 3:227, 3:227, :].transpose(0, 3, 1, 2), rgb_ref)
    border = r3.copy()
    border[:, 3:227, 3:227, :] = 0
    assert (border == 0).all()
    with pytest.raises(RuntimeError):
        pp(torch.from_numpy(frames))


@pytest.mark.gpu
def test_gpu_preprocessing_word_and_byte_forms_agree(pkg):
    """Round 6: preproc_gray_words_kernel / preproc_rgb3_words_kernel (every LDS access a word or wider: profiles/r06_ab_preproc_words.txt) need
    4-byte aligned frames and fall back to the byte-wide kernels of rounds 2-5 otherwise.  The same frames at an aligned and at an odd
    device address must give the same bits -- both equal to PIL -- so the launcher's choice changes no result."""
    import torch
    from mimamo_net_amd.preprocess import FramePreprocessor
    dev = torch.device("cuda:0")
    frames = np.concatenate([synthetic.make_clip_u8(5, 5), (weights.det_uniform("pp.noise2", (4, 112, 112, 3), 0, 256, 9)).astype(np.uint8)])
    gray_ref, rgb_ref = synthetic.preprocess_host(frames)
    pp = FramePreprocessor(device=dev)
    aligned = torch.from_numpy(frames).to(dev)
    raw = torch.empty(frames.size + 3, dtype=torch.uint8, device=dev)
    outs = []
    for shift in (0, 1, 2, 3):
        view = raw[shift:shift + frames.size].view(frames.shape)
        view.copy_(aligned)
        assert view.data_ptr() % 4 == (raw.data_ptr() + shift) % 4
        g, r3 = pp(view, bordered3=True)
        outs.append((g.cpu().numpy(), r3.cpu().numpy()))
    assert {(raw.data_ptr() + s) % 4 for s in range(4)} == {0, 1, 2, 3}         # one aligned view (word kernels), three odd ones (byte kernels)
    for g, r3 in outs:
        np.testing.assert_array_equal(g, gray_ref)
        np.testing.assert_array_equal(r3[:, 3:227, 3:227, :].transpose(0, 3, 1, 2), rgb_ref)
        np.testing.assert_array_equal(r3, outs[0][1])
