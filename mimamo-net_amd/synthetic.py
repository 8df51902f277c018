"""Synthetic aligned-face clips (bench / parity inputs).

The reference's real inputs are OpenFace crops of a video (api/video_processor.py:69-84):
112x112 RGB uint8 BMPs `frame_det_00_%06d.bmp`.  OpenFace, ffmpeg and the example weights
are not available offline, so the bench uses a stand-in of the same shape and statistics:
textured (never constant -- the pyramid phase is undefined where the magnitude vanishes)
and temporally coherent (a drifting low-pass noise field + 2 % white noise), seeded per
clip with `1000 + clip_id`.  Everything is closed-form numpy so the
same bytes are produced on every host.
"""
import numpy as np

from .weights import det_uniform

FRAME = 112


def make_clip_u8(clip_id, n_frames=64, size=FRAME):
    """[n_frames, size, size, 3] uint8."""
    seed = 1000 + int(clip_id)
    fy = np.fft.fftfreq(size)[:, None]
    fx = np.fft.fftfreq(size)[None, :]
    lowpass = np.exp(-(fx ** 2 + fy ** 2) / (2 * 0.045 ** 2))
    base = []
    for ch in range(3):
        noise = det_uniform("clip.base.%d" % ch, (size, size), -1.0, 1.0, seed).astype(np.float64)
        spec = np.fft.fft2(noise) * lowpass
        base.append(spec)
    out = np.empty((n_frames, size, size, 3), dtype=np.uint8)
    for t in range(n_frames):
        shift = np.exp(-2j * np.pi * (fy * 0.3 * t + fx * 0.2 * t))
        wn = det_uniform("clip.noise", (size, size, 3), -1.0, 1.0, seed * 131 + t).astype(np.float64)
        for ch in range(3):
            img = np.real(np.fft.ifft2(base[ch] * shift))
            img = img / (np.abs(img).max() + 1e-12)
            img = 0.5 + 0.42 * img + 0.02 * wn[..., ch]
            out[t, ..., ch] = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
    return out


def to_gray_u8(frames_u8):
    """PIL `convert('L')` (snippet_sampler.py:163): (19595 R + 38470 G + 7471 B + 0x8000) >> 16."""
    f = frames_u8.astype(np.uint32)
    return ((19595 * f[..., 0] + 38470 * f[..., 1] + 7471 * f[..., 2] + 0x8000) >> 16).astype(np.uint8)


def preprocess_host(frames_u8, phase_size=48, mean=(131.0912, 103.8827, 91.4953)):
    """Host (PIL) version of the reference's two image transforms, used to prepare inputs
    until the on-GPU preprocessing row replaces it.

    gray: convert('L') -> Lanczos resize to phase_size -> /255   (snippet_sampler.py:163,177-185)
    rgb : Resize(256, bilinear) -> CenterCrop(224) -> ToTensor -> x255 -> -mean
          (utils/model_utils.py:29-39)
    Returns (gray [N,48,48] f32, rgb [N,3,224,224] f32).
    """
    from PIL import Image
    n = frames_u8.shape[0]
    gray = np.empty((n, phase_size, phase_size), dtype=np.float32)
    rgb = np.empty((n, 3, 224, 224), dtype=np.float32)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    for i in range(n):
        im = Image.fromarray(frames_u8[i], "RGB")
        g = im.convert("L").resize((phase_size, phase_size), Image.LANCZOS)
        gray[i] = np.asarray(g, dtype=np.float32) / np.float32(255)
        r = np.asarray(im.resize((256, 256), Image.BILINEAR), dtype=np.uint8)[16:240, 16:240]
        x = (r.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)
        rgb[i] = x * np.float32(255.0) - m
    return gray, rgb


def textured_gray(n, size=48, seed=7):
    """Cheap textured [n,size,size] f32 in [0,1] on the k/255 lattice, drifting over n."""
    fy = np.fft.fftfreq(size)[:, None]
    fx = np.fft.fftfreq(size)[None, :]
    lowpass = np.exp(-(fx ** 2 + fy ** 2) / (2 * 0.09 ** 2))
    spec = np.fft.fft2(det_uniform("gray.base", (size, size), -1, 1, seed).astype(np.float64)) * lowpass
    out = np.empty((n, size, size), dtype=np.float32)
    for t in range(n):
        shift = np.exp(-2j * np.pi * (fy * 0.37 * t + fx * 0.23 * t))
        img = np.real(np.fft.ifft2(spec * shift))
        img = 0.5 + 0.42 * img / (np.abs(img).max() + 1e-12)
        img = img + 0.02 * det_uniform("gray.noise", (size, size), -1, 1, seed * 977 + t)
        out[t] = (np.clip(np.rint(img * 255.0), 0, 255) / 255.0).astype(np.float32)
    return out
