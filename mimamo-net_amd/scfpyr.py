"""SCFpyr_PyTorch on MI355X -- drop-in for api/steerable/SCFpyr_PyTorch.py:51-208 (construction only).

`build(im_batch)` returns the reference's full list `[hi, [band_0..band_{nbands-1}], ..., lo]` for arbitrary
square images; the arithmetic runs in libmimamo_hip.so (csrc/scfpyr.hip: DFT-by-summation with float64
accumulation).  The inference pipeline does not use this class -- Phase_Difference_Extractor.build_pyramid calls
the mirrored-input kernel of csrc/pyramid.hip, which produces only the coefficients the phase stage keeps.
`reconstruct` (SCFpyr_PyTorch.py:213-318) is never reached by inference and is not implemented.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class SCFpyr_PyTorch(object):
    def __init__(self, height=5, nbands=4, scale_factor=2, device=None, precision=32):
        """Arguments as SCFpyr_PyTorch.py:51-58.  Unlike the reference this does NOT call
        torch.set_default_dtype (quirk Q10) and the device must be a ROCm device."""
        self.height = height
        self.nbands = nbands
        self.scale_factor = scale_factor
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("SCFpyr_PyTorch: this build has no CPU path; pass a ROCm device")
        self.precision = precision
        assert self.precision in [32, 64]
        self.dtype = torch.float32 if precision == 32 else torch.float64
        self._handle = None
        self._size = None

    def _get(self, size):
        if self._handle is None or self._size != size:
            self.close()
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                rc = _lib.lib().mm_scfpyr_create(ctypes.byref(h), int(size), int(self.height), int(self.nbands),
                                                 int(self.scale_factor))
            if rc == _lib.MM_ERR_TOO_SMALL:  # the reference formats the level count into the message (:91)
                raise RuntimeError('Cannot build {} levels, image too small.'.format(self.height))
            _lib.check(rc, "mm_scfpyr_create")
            self._handle, self._size = h, size
        return self._handle

    def close(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().mm_scfpyr_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, im_batch):
        """im_batch [N,1,H,W] -> [hi [N,H,W], [bands [N,h,w,2]] per level ..., lo [N,h',w']]  (SCFpyr_PyTorch.py:70-125)."""
        assert im_batch.device == self.device, 'Devices invalid (pyr = {}, batch = {})'.format(self.device, im_batch.device)
        assert im_batch.dtype == self.dtype, 'Image batch must be torch.float{}'.format(self.precision)
        assert im_batch.dim() == 4, 'Image batch must be of shape [N,C,H,W]'
        assert im_batch.shape[1] == 1, 'Second dimension must be 1 encoding grayscale image'
        n, _, hh, ww = im_batch.shape
        if hh != ww:
            raise NotImplementedError("square images only (SCFpyr_PyTorch.py:87 swaps height and width)")
        h = self._get(hh)
        L = _lib.lib()
        n_out = L.mm_scfpyr_num_outputs(h)
        outs = []
        side, cplx = ctypes.c_int(), ctypes.c_int()
        for i in range(n_out):
            _lib.check(L.mm_scfpyr_output_info(h, i, ctypes.byref(side), ctypes.byref(cplx)), "mm_scfpyr_output_info")
            shape = (n, side.value, side.value, 2) if cplx.value else (n, side.value, side.value)
            outs.append(torch.empty(shape, dtype=self.dtype, device=self.device))
        ws_bytes = L.mm_scfpyr_workspace_bytes(h, n)
        ws = torch.empty((max(ws_bytes, 8) // 8,), dtype=torch.float64, device=self.device)
        ptrs = (ctypes.c_void_p * n_out)(*[o.data_ptr() for o in outs])
        x = im_batch.contiguous()
        with torch.cuda.device(self.device):
            rc = L.mm_scfpyr_build(h, _lib.ptr(x), self.precision, n, ptrs, _lib.ptr(ws), ws_bytes, _lib.current_stream())
        _lib.check(rc, "mm_scfpyr_build")
        coeff = [outs[0]]
        k = 1
        for _ in range(self.height - 2):
            coeff.append(outs[k:k + self.nbands])
            k += self.nbands
        coeff.append(outs[k])
        return coeff
