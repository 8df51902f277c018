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


# ---- drop-in Dataset classes (reference names and item layouts) ----------------------------------------------------------------
class _VideoRecord(object):
    """Frame list + dummy labels of one processed video (api/sampler/snippet_sampler.py:12-63, test mode)."""

    def __init__(self, video, frames, label_name):
        self.video = video
        names = None if label_name is None else ([label_name] if '_' not in label_name else label_name.split("_"))
        self.label_name = names
        if len(frames) == 0:
            raise ValueError("number of frames of video {} should not be zero.".format(video))
        n_lab = 1 if names is None else len(names)
        self.path_label = [frames, np.array([[-100] * n_lab] * len(frames))]


def _need_test_mode(test_mode, who):
    if not test_mode:
        # training-time sampling (annotation files, random flip / crop / colour jitter) is outside the inference hot path
        # (Aff-wild-exps / OMG-exps are out of scope); api/'s own non-test branch of Snippet_Sampler reads
        # `annot_file` before assigning it (snippet_sampler.py:33-35) and cannot run there either
        raise NotImplementedError("%s(test_mode=False): training-time sampling is outside this build" % who)


class Snippet_Sampler(object):
    """Drop-in for api/sampler/snippet_sampler.py:66-186 (a map-style dataset: usable with torch.utils.data.DataLoader).

    Same constructor keywords, same `seq_ranges`, same item: `(phase_images [T, num_phase+1, S, S] float32 tensor, rgb features
    [T, 2048], dummy labels [T, n], np.array([start, end]), video name)` -- `Tester.test_on_dataloader` consumes it unchanged.
    Differences, both on the host side only: every BMP of a snippet is decoded ONCE (the reference re-opens each frame for every
    window that contains it, up to 13 times, :156-167), and with `return_u8=True` the item carries the raw boundary instead of
    the PIL-preprocessed windows: `(frames_u8 [U, h, w, 3] uint8 -- the unique frames of the snippet's windows --, window ids
    [T, num_phase+1] int32 into them, rgb features, labels, range, name)`, to be resized on the GPU (FramePreprocessor, bit-exact
    with the PIL path) and fed to Phase_Difference_Extractor.phase_diff_frames."""

    def __init__(self, video_name, root_path, feature_path, annot_dir=None, label_name=None, test_mode=True,
                 num_phase=12, phase_size=48, length=64, stride=64, verbose=False, return_u8=False):
        _need_test_mode(test_mode, "Snippet_Sampler")
        self.video_name, self.root_path, self.feature_path, self.annot_dir = video_name, root_path, feature_path, annot_dir
        self.label_name, self.test_mode = label_name, test_mode
        self.length, self.stride, self.num_phase, self.phase_size = length, stride, num_phase, phase_size
        self.verbose, self.return_u8 = verbose, return_u8
        self.parse_video()

    def parse_video(self):
        frames = glob.glob(os.path.join(self.feature_path, '*.npy'))
        frames = sorted(frames, key=lambda x: os.path.basename(x).split(".")[0])          # :24-25
        self.video_record = _VideoRecord(self.video_name, frames, self.label_name)
        if len(frames) < self.length:
            print("The length exceeds the number of exsisting frames, the sampling length has been changed to {}".format(len(frames)))
            self.length = self.stride = len(frames)
        self.seq_ranges = snippet_ranges(len(frames), self.length, self.stride)
        if self.verbose:
            print("videos {}, number of seqs:{}".format(self.video_name, len(self.seq_ranges)))

    def __len__(self):
        return len(self.seq_ranges)

    def _face_path(self, feature_file):
        f_index = int(os.path.basename(feature_file).split(".")[0])
        return os.path.join(self.root_path, self.video_record.video + "_aligned", 'frame_det_00_{:06d}.bmp'.format(f_index))

    def __getitem__(self, index):
        import torch
        from PIL import Image
        start, end = self.seq_ranges[index]
        frames, labels = self.video_record.path_label
        imgs = np.array([np.load(f) for f in frames[start:end]])
        ids = window_ids(start, end, len(frames), self.num_phase)                          # [T, num_phase + 1], clamped (:144-152)
        uniq, inv = np.unique(ids, return_inverse=True)
        inv = inv.reshape(ids.shape)
        try:
            faces = [Image.open(self._face_path(frames[i])) for i in uniq]
            faces = [f.convert('RGB' if self.return_u8 else 'L') for f in faces]
        except Exception:
            raise ValueError("incorrect face path")
        rng = np.array([start, end])
        if self.return_u8:
            u8 = np.stack([np.asarray(f, dtype=np.uint8) for f in faces])
            return torch.from_numpy(u8), inv.astype(np.int32), imgs, np.array(labels[start:end]), rng, self.video_record.video
        S = self.phase_size
        planes = np.empty((len(uniq), S, S), dtype=np.float32)
        for k, g in enumerate(faces):                                                       # GroupScale(size) + /255 (:177-185)
            w, h = g.size
            if not ((w <= h and w == S) or (h <= w and h == S)):
                ow, oh = (S, int(S * h / w)) if w < h else (int(S * w / h), S)
                g = g.resize((ow, oh), Image.LANCZOS)
            if g.size != (S, S):
                raise ValueError("aligned face is %dx%d: not square, cannot be viewed as %dx%d phase input" % (w, h, S, S))
            planes[k] = np.asarray(g, dtype=np.float32) / np.float32(255)
        phase_images = torch.from_numpy(planes[inv])                                        # [T, num_phase + 1, S, S]
        return phase_images, imgs, np.array(labels[start:end]), rng, self.video_record.video


class Image_Sampler(object):
    """Drop-in for api/sampler/image_sampler.py:64-143: one aligned face per item, `(image, label, frame path, video name)`.

    `transform` is applied to the PIL image as in the reference (Resnet50_Extractor passes the model's own, api/
    resnet50_extractor.py:53).  With `return_u8=True` the image is the decoded uint8 frame [h, w, 3] (no transform): the raw
    boundary the GPU preprocessing takes (FramePreprocessor: bilinear 256 + centre crop 224 + 255 x - mean, bit-exact with PIL).
    Without a transform in test mode the reference builds Resize(size) + ToTensor + ImageNet Normalize (:131-138): reproduced with
    PIL + numpy (torchvision is not needed)."""

    def __init__(self, video_name, root_path, test_mode=False, annot_dir=None, label_name=None, transform=None, verbose=False,
                 size=224, return_u8=False):
        _need_test_mode(test_mode, "Image_Sampler")
        self.video_name, self.root_path, self.annot_dir, self.label_name = video_name, root_path, annot_dir, label_name
        self.test_mode, self.transform, self.size, self.verbose, self.return_u8 = test_mode, transform, size, verbose, return_u8
        self.parse_video()

    def parse_video(self):
        frames = glob.glob(os.path.join(self.root_path, self.video_name + "_aligned", '*.bmp'))
        frames = sorted(frames, key=lambda x: os.path.basename(x).split(".")[0].split("_")[-1])   # :26-27
        self.video_record = _VideoRecord(self.video_name, frames, self.label_name)
        if self.verbose:
            print("video {} has {} frames".format(self.video_name, len(frames)))
        self.frame_ids = np.arange(len(frames))

    def __len__(self):
        return len(self.frame_ids)

    def _default_transform(self, img):
        import torch
        from PIL import Image
        w, h = img.size
        s = self.size
        if not ((w <= h and w == s) or (h <= w and h == s)):
            img = img.resize((s, int(s * h / w)) if w < h else (int(s * w / h), s), Image.BILINEAR)
        a = np.asarray(img.convert('RGB'), dtype=np.float32) / np.float32(255)
        a = (a - np.asarray([0.485, 0.456, 0.406], dtype=np.float32)) / np.asarray([0.229, 0.224, 0.225], dtype=np.float32)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    def __getitem__(self, index):
        import torch
        from PIL import Image
        f_id = self.frame_ids[index]
        frames, labels = self.video_record.path_label
        frame, label = frames[f_id], labels[f_id]
        img = Image.open(frame)
        if self.return_u8:
            img = torch.from_numpy(np.asarray(img.convert('RGB'), dtype=np.uint8).copy())
        else:
            img = self.transform(img) if self.transform is not None else self._default_transform(img)
        return img, label, frame, self.video_record.video
