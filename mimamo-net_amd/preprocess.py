"""On-GPU frame preprocessing, bit-exact with the PIL transforms of the reference's samplers
(api/sampler/snippet_sampler.py:163,177-185 + api/utils/data_utils.py:80; api/utils/model_utils.py:29-39):
uint8 aligned faces [n,112,112,3] -> gray [n,48,48] f32 and RGB [n,3,224,224] (or NHWC4) f32 = 255*x - mean."""
import ctypes

import torch

from . import _lib
from .weights import RESNET50_MEAN


class FramePreprocessor(object):
    def __init__(self, in_size=112, phase_size=48, resize=256, crop=224, mean=RESNET50_MEAN, device=None):
        self.in_size, self.phase_size, self.resize, self.crop = in_size, phase_size, resize, crop
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        m = (ctypes.c_float * 3)(*[float(v) for v in mean])
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = _lib.lib().mm_preproc_create(ctypes.byref(h), in_size, phase_size, resize, crop, m)
        _lib.check(rc, "mm_preproc_create")
        self._handle = h

    def close(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().mm_preproc_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, frames_u8, channels_last4=True, want_gray=True, want_rgb=True, bordered3=False):
        """frames_u8: uint8 device tensor [n,S,S,3].  Returns (gray [n,48,48] or None, rgb or None).
        rgb layout: [n,3,crop,crop] (channels_last4=False), [n,crop,crop,4] (channels_last4=True) or, with bordered3=True, packed
        three-channel rows with the stem's zero border in memory [n,crop+6,crop+6,3] (Resnet50_Extractor's fastest input)."""
        if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8:
            raise RuntimeError("frames must be a uint8 tensor on the ROCm device")
        n = frames_u8.shape[0]
        assert tuple(frames_u8.shape[1:]) == (self.in_size, self.in_size, 3)
        frames_u8 = frames_u8.contiguous()
        gray = torch.empty((n, self.phase_size, self.phase_size), dtype=torch.float32, device=frames_u8.device) if want_gray else None
        if want_rgb:
            shape = ((n, self.crop + 6, self.crop + 6, 3) if bordered3 else
                     (n, self.crop, self.crop, 4) if channels_last4 else (n, 3, self.crop, self.crop))
            rgb = torch.empty(shape, dtype=torch.float32, device=frames_u8.device)
        else:
            rgb = None
        rc = _lib.lib().mm_preproc_forward(self._handle, _lib.ptr(frames_u8), n, _lib.ptr(gray), _lib.ptr(rgb),
                                           2 if bordered3 else 0 if channels_last4 else 1, _lib.current_stream())
        _lib.check(rc, "mm_preproc_forward")
        return gray, rgb
