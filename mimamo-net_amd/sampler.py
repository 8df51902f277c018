"""Snippet / window index logic, frame IO and result assembly (host side).

Mirrors api/sampler/snippet_sampler.py:107-152 (snippet ranges, 13-frame clamped windows),
api/sampler/image_sampler.py + api/utils/model_utils.py:26-40 (RGB transform), api/tester.py:94-121
(assembly).  Pure index/IO code: no torch DataLoader worker processes, no per-window re-opening of
the same BMP (the reference opens every frame up to 13 times, snippet_sampler.py:156-167).
"""
import glob
import os

import numpy as np

from .weights import RESNET50_MEAN


def snippet_ranges(n_frames, length=64, stride=64):
    """[[start, end], ...] incl. the short-video rule and the tail snippet (snippet_sampler.py:112-126)."""
    if n_frames <= 0:
        raise ValueError("number of frames of video should not be zero.")  # snippet_sampler.py:28
    if n_frames < length:
        length = stride = n_frames
    ranges = []
    start, end = 0, length
    while end <= n_frames and start < n_frames:
        ranges.append([start, end])
        start += stride
        end = start + length
    assert len(ranges) != 0, "No snippet is sampled."
    if ranges[-1][1] < n_frames:
        ranges.append([n_frames - length, n_frames])
    return ranges


def window_ids(start, end, n_frames, num_phase=12):
    """int32 [end-start, num_phase+1]: clamp(f + i - num_phase//2, 0, n_frames-1) (snippet_sampler.py:144-152)."""
    f = np.arange(start, end, dtype=np.int64)[:, None] + (np.arange(num_phase + 1, dtype=np.int64)[None, :] - num_phase // 2)
    return np.clip(f, 0, n_frames - 1).astype(np.int32)


def assemble(preds, ranges, n_labels=2):
    """video[start:end] = pred in sampler order; later snippets overwrite (tester.py:103-118)."""
    max_len = max(r[1] for r in ranges)
    video = np.zeros((max_len, n_labels))
    lo, hi = 0, 0
    for (s, e), p in zip(ranges, preds):
        video[s:e, :] = p
        lo, hi = min(lo, s), max(hi, e)
    assert (lo == 0) and (hi == max_len)
    return video


# ---- on-disk formats of the reference ------------------------------------------------
def list_aligned_frames(opface_dir, video_name):
    """[(1-based frame index, path)] of `<opface_dir>/<video>_aligned/frame_det_00_%06d.bmp`, sorted."""
    d = os.path.join(opface_dir, video_name + "_aligned")
    paths = glob.glob(os.path.join(d, "frame_det_00_*.bmp"))
    out = sorted((int(os.path.basename(p).split('.')[0].split('_')[-1]), p) for p in paths)
    if not out:
        raise ValueError("no aligned faces under %s" % d)
    return out


def load_u8_batch(paths, size=112):
    """Decoded RGB frames as one uint8 array [n,size,size,3], or None when a frame has another size (then the
    host-side PIL preprocessing below applies)."""
    from PIL import Image
    out = np.empty((len(paths), size, size, 3), dtype=np.uint8)
    for i, p in enumerate(paths):
        im = Image.open(p).convert('RGB')
        if im.size != (size, size):
            return None
        out[i] = np.asarray(im, dtype=np.uint8)
    return out


def load_rgb_batch(paths, mean=RESNET50_MEAN):
    """BMP -> Resize(256, bilinear) -> CenterCrop(224) -> ToTensor -> x255 -> -mean (utils/model_utils.py:29-39).
    Returns a CPU float tensor [n,3,224,224]."""
    import torch
    from PIL import Image
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    out = np.empty((len(paths), 3, 224, 224), dtype=np.float32)
    for i, p in enumerate(paths):
        im = Image.open(p).convert('RGB')
        w, h = im.size
        if w <= h:
            ow, oh = 256, int(256 * h / w)
        else:
            oh, ow = 256, int(256 * w / h)
        im = im.resize((ow, oh), Image.BILINEAR)
        left, top = int(round((ow - 224) / 2.0)), int(round((oh - 224) / 2.0))
        a = np.asarray(im.crop((left, top, left + 224, top + 224)), dtype=np.uint8)
        out[i] = (a.astype(np.float32) / np.float32(255)).transpose(2, 0, 1) * np.float32(255.0) - m
    return torch.from_numpy(out)


def load_gray_batch(paths, phase_size=48):
    """BMP -> convert('L') -> Lanczos resize -> /255 (snippet_sampler.py:163,177-185).  CPU tensor [n,S,S]."""
    import torch
    from PIL import Image
    out = np.empty((len(paths), phase_size, phase_size), dtype=np.float32)
    for i, p in enumerate(paths):
        g = Image.open(p).convert('L')
        w, h = g.size
        if w <= h:
            ow, oh = phase_size, int(phase_size * h / w)
        else:
            oh, ow = phase_size, int(phase_size * w / h)
        if (ow, oh) != (phase_size, phase_size):
            # the reference's `.view(len, 13, 48, 48)` (snippet_sampler.py:185) raises for non-square faces; a silent
            # top-left crop would hand plausible but shifted phase images to the network instead
            raise ValueError("aligned face %s is %dx%d: not square, cannot be viewed as %dx%d phase input"
                             % (p, w, h, phase_size, phase_size))
        g = g.resize((ow, oh), Image.LANCZOS)
        out[i] = np.asarray(g, dtype=np.float32) / np.float32(255)
    return torch.from_numpy(out)
