"""Phase_Difference_Extractor on MI355X -- drop-in for api/phase_difference_extractor.py:6-134.

Same constructor arguments, same method names, same tensor layouts; the arithmetic runs in
libmimamo_hip.so (csrc/pyramid.hip, csrc/phase_window.hip) through the C ABI.  Tensors must live
on a ROCm device ("cuda"); there is no CPU fallback.
"""
import ctypes
import weakref

import torch

from . import _lib


class Phase_Difference_Extractor(object):
    def __init__(self, height=5, nbands=4, scale_factor=2, extract_level=1, visualize=False):
        """Arguments as api/phase_difference_extractor.py:7-37.  The configuration the reference's Tester uses
        (api/tester.py:28-32: height=4, nbands=2, scale_factor=2, levels 1/2, 48x48 frames, 13-frame windows) runs on
        the fused hot-path kernels; every other configuration goes through the general pyramid (csrc/scfpyr.hip,
        square frames of any parity up to 512x512 with symmetry, 1024x1024 without) and the generic extract kernel
        (csrc/phase_generic.hip: planes up to 4096 pixels in LDS, larger ones through a workspace).
        RuntimeError 'image too small' keeps the reference's meaning (SCFpyr_PyTorch.py:90-91)."""
        if visualize:
            raise NotImplementedError("visualize=True is a debug path of the reference (matplotlib); out of scope")
        self.height = height
        self.nbands = nbands
        self.scale_factor = scale_factor
        self.extract_level = extract_level
        self.visualize = visualize
        self._handle = None
        self._size = None
        self._ids_cache = {}
        self._ws = {}          # per-stream workspace of the fused path (pyramid planes, 46 KB per unique frame)
        self._ids_ok = {}      # window-id tables already range-checked: id(tensor) -> (weakref, (version, N))
        self._general = None   # SCFpyr_PyTorch for configurations outside the fused kernels

    def _levels(self):
        return [self.extract_level] if isinstance(self.extract_level, int) else list(self.extract_level)

    def _fused(self, W, symmetry=True):
        """True when the hot-path kernels (pyramid.hip / phase_window.hip) implement this configuration."""
        return (symmetry and W == 48 and (self.height, self.nbands, self.scale_factor) == (4, 2, 2)
                and all(lv in (1, 2) for lv in self._levels()))

    # -- native handle -------------------------------------------------------------------
    def _get(self, size):
        if self._handle is None or self._size != size:
            self.close()
            h = ctypes.c_void_p()
            rc = _lib.lib().mm_pyramid_create(ctypes.byref(h), int(size), int(self.height), int(self.nbands),
                                              int(self.scale_factor))
            _lib.check(rc, "mm_pyramid_create")
            self._handle, self._size = h, size
        return self._handle

    def close(self):
        if self._handle is not None:
            _lib.lib().mm_pyramid_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check_input(t, ndim, what):
        if not isinstance(t, torch.Tensor):
            raise ValueError("%s must be a torch.Tensor" % what)
        if t.dim() != ndim:
            raise ValueError("%s must have %d dimensions, got %d" % (what, ndim, t.dim()))
        if not t.is_cuda:
            raise RuntimeError("%s must be on a ROCm device (cuda); this build has no CPU path" % what)
        assert t.dtype == torch.float32, "Image batch must be torch.float32"  # SCFpyr_PyTorch.py:82

    # -- reference API -------------------------------------------------------------------
    def build_pyramid(self, im_batch, symmetry=True):
        """im_batch [B, P, W, H] -> coefficients [B, nbands, P, W_l, H_l, 2] (a list when extract_level is
        a list) -- api/phase_difference_extractor.py:38-87."""
        self._check_input(im_batch, 4, "im_batch")
        B, P, W, H = im_batch.shape
        assert W == H, "square frames only (SCFpyr_PyTorch.py:87 swaps height/width)"
        if not self._fused(W, symmetry):
            return self._build_pyramid_general(im_batch.contiguous(), symmetry)
        h = self._get(W)
        levels = self._levels()
        x = im_batch.contiguous()
        c1 = torch.empty((B, self.nbands, P, W, H, 2), dtype=torch.float32, device=x.device)
        c2 = torch.empty((B, self.nbands, P, W // 2, H // 2, 2), dtype=torch.float32, device=x.device)
        rc = _lib.lib().mm_pyramid_build_batch(h, _lib.ptr(x), B, P, _lib.ptr(c1), _lib.ptr(c2), _lib.current_stream())
        _lib.check(rc, "mm_pyramid_build_batch")
        out = [c1 if lv == 1 else c2 for lv in levels]
        return out[0] if isinstance(self.extract_level, int) else out

    def _build_pyramid_general(self, x, symmetry):
        """Any other configuration: mirror (index-only, torch ops), the general steerable pyramid, then the reference's
        stack / view / permute / quadrant crop (:71-86)."""
        from .scfpyr import SCFpyr_PyTorch
        B, P, W, H = x.shape
        im = x.view(B * P, 1, W, H)
        if symmetry:   # symmetric_extension_batch, api/utils/phase_utils.py:116-129
            top = torch.cat([im, im.flip(-1)], dim=-1)
            im = torch.cat([top, top.flip(-2)], dim=-2).contiguous()
        if self._general is None or self._general.device != x.device:
            self._general = SCFpyr_PyTorch(self.height, self.nbands, self.scale_factor, device=x.device, precision=32)
        coeff = self._general.build(im)
        outs = []
        for lv in self._levels():
            bands = coeff[lv]
            assert isinstance(bands, list)   # extract_coeff_level (:90): residual levels are not lists
            c = torch.stack(bands, 0)         # [nbands, B*P, w, h, 2]
            w, h = c.shape[-3], c.shape[-2]
            c = c.view(len(bands), B, P, w, h, 2).permute(1, 0, 2, 3, 4, 5).contiguous()
            if symmetry:
                c = c[..., : w // 2, : h // 2, :]
            outs.append(c)
        return outs[0] if isinstance(self.extract_level, int) else outs

    def extract(self, coeff_batch):
        """coeff [B, nbands, P, W, H, 2] -> phase differences [B, nbands, P-1, W, H]
        (api/phase_difference_extractor.py:93-134)."""
        if isinstance(coeff_batch, (list, tuple)):
            raise ValueError("extract() takes the coefficients of ONE level")
        self._check_input(coeff_batch, 6, "coeff_batch")
        B, nb, P, W, H, two = coeff_batch.shape
        assert two == 2
        c = coeff_batch.contiguous()
        out = torch.empty((B, nb, P - 1, W, H), dtype=torch.float32, device=c.device)
        if not (W == H and W in (48, 24) and nb == 2 and P == 13 and (self.height, self.nbands, self.scale_factor) == (4, 2, 2)):
            self._extract_generic(c, B * nb, P, W, H, out, None)
            return out
        h = self._get(self._size if self._size is not None else 48)
        key = (B, P, nb, str(c.device))
        ids = self._ids_cache.get(key)
        if ids is None:
            # plane (b, band, p) sits at ((b*nb + band)*P + p) * plane: ids = b*nb*P + p, band stride P*plane
            ids = (torch.arange(B, device=c.device, dtype=torch.int32)[:, None] * (nb * P)
                   + torch.arange(P, device=c.device, dtype=torch.int32)[None, :]).contiguous()
            self._ids_cache = {key: ids}
        plane = W * H * 2
        rc = _lib.lib().mm_phase_extract(h, _lib.ptr(c), _lib.ptr(ids), plane, P * plane, B, P, W, _lib.ptr(out),
                                         0, 0, 0, _lib.current_stream())
        _lib.check(rc, "mm_phase_extract")
        return out

    @staticmethod
    def _extract_generic(c, planes, P, W, H, out, denoised, force_workspace=False):
        """The generic extract kernel; planes above 4 096 pixels (or force_workspace, for tests) take the workspace form
        (mm_phase_extract_generic_ws: 8 floats per pixel and plane set, allocated per call)."""
        L = _lib.lib()
        need = L.mm_phase_extract_generic_workspace_bytes(planes, P, W, H)
        if need < 0:
            _lib.check(int(need), "mm_phase_extract_generic_workspace_bytes")
        if force_workspace and need == 0:
            need = planes * W * H * 8 * 4
        if need == 0:
            rc = L.mm_phase_extract_generic(_lib.ptr(c), planes, P, W, H, _lib.ptr(out), _lib.ptr(denoised), _lib.current_stream())
            _lib.check(rc, "mm_phase_extract_generic")
            return
        ws = torch.empty((need // 4,), dtype=torch.float32, device=c.device)
        rc = L.mm_phase_extract_generic_ws(_lib.ptr(c), planes, P, W, H, _lib.ptr(out), _lib.ptr(denoised), _lib.ptr(ws), need,
                                           _lib.current_stream())
        _lib.check(rc, "mm_phase_extract_generic_ws")

    # -- fused fast path (not in the reference API) ---------------------------------------
    def phase_diff_frames(self, frames, window_ids, nhwc=False, out1_cstride=None, out1_coffset=0, ids_checked=False):
        """De-duplicated driver: unique frames [N, W, W] + window ids [J, 13] (int32, clamped frame indices,
        api/sampler/snippet_sampler.py:144-152) -> (phase_0 [J,24,W,W], phase_1 [J,24,W/2,W/2]).

        Builds each frame's pyramid once instead of once per window that contains it (13x less work than
        Tester.phase_diff_output; quirk Q3) and blurs everything that is linear in the frame once per frame: equal to the
        windowed evaluation up to fp32 rounding of the regrouping B + blur(mag * acc) * R (csrc/phase_frames.hip; tests:
        <= 5e-5 against the literal kernel).
        nhwc=True writes channels-last tensors ([J,W,W,24], [J,W/2,W/2,out1_cstride] with the 24 channels
        at out1_coffset) for the head's conv engine.
        The ids index the N frames handed over; any pattern is allowed (the unwrap decisions are taken between the
        consecutive frames of each window).  The range is checked on the host the first time a table is seen (one
        device->host read, cached per table); ids_checked=True skips that for tables the caller built from the same
        frame count (HotPath.plan)."""
        self._check_input(frames, 3, "frames")
        N, W, _ = frames.shape
        J = window_ids.shape[0]
        assert window_ids.dtype == torch.int32 and window_ids.is_cuda and tuple(window_ids.shape) == (J, 13)
        if not ids_checked and J > 0:
            hit = self._ids_ok.get(id(window_ids))
            if not (hit is not None and hit[0]() is window_ids and hit[1] == (window_ids._version, N)):
                chk = torch.stack([window_ids.min(), window_ids.max()]).tolist()   # one device->host read
                lo, hi = int(chk[0]), int(chk[1])
                if lo < 0 or hi >= N:
                    raise ValueError("window_ids must index the %d frames handed over (found %d..%d)" % (N, lo, hi))
                if len(self._ids_ok) > 64:
                    self._ids_ok.clear()
                self._ids_ok[id(window_ids)] = (weakref.ref(window_ids), (window_ids._version, N))
        h = self._get(W)
        L = _lib.lib()
        ws_bytes = L.mm_phase_workspace_bytes(h, N)
        skey = torch.cuda.current_stream().cuda_stream
        ws = self._ws.get(skey)
        if ws is None or ws.numel() * 4 < ws_bytes or ws.device != frames.device:
            self._ws[skey] = None
            if len(self._ws) > 8:
                self._ws = {}
            ws = self._ws[skey] = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=frames.device)
        C = 2 * 12
        if nhwc:
            cs1 = C if out1_cstride is None else out1_cstride
            p0 = torch.empty((J, W, W, C), dtype=torch.float32, device=frames.device)
            p1 = torch.empty((J, W // 2, W // 2, cs1), dtype=torch.float32, device=frames.device)
            args = (_lib.ptr(p0), 1, C, 0, _lib.ptr(p1), 1, cs1, out1_coffset)
        else:
            p0 = torch.empty((J, C, W, W), dtype=torch.float32, device=frames.device)
            p1 = torch.empty((J, C, W // 2, W // 2), dtype=torch.float32, device=frames.device)
            args = (_lib.ptr(p0), 0, 0, 0, _lib.ptr(p1), 0, 0, 0)
        rc = L.mm_phase_diff_frames(h, _lib.ptr(frames.contiguous()), N, _lib.ptr(window_ids.contiguous()), J,
                                    *args, _lib.ptr(ws), ws_bytes, _lib.current_stream())
        _lib.check(rc, "mm_phase_diff_frames")
        return p0, p1


    def phase_diff_planes(self, planes, window_ids, out=None):
        """The window half of phase_diff_frames on per-frame planes the caller holds: planes [N, 2, 4, W, W] =
        {magnitude, B = blur(mag phase)/blur(mag), R = 1/blur(mag), wrapped phase} per (frame, band), W = 48 or 24
        (what mm_phase_diff_frames leaves in its workspace); window_ids int32 [J, 13] -> [J, 24, W, W]."""
        self._check_input(planes, 5, "planes")
        N, nb, four, W, _ = planes.shape
        assert nb == 2 and four == 4 and window_ids.dtype == torch.int32 and window_ids.is_cuda
        J = window_ids.shape[0]
        if out is None:
            out = torch.empty((J, 24, W, W), dtype=torch.float32, device=planes.device)
        rc = _lib.lib().mm_phase_diff_planes(self._get(48), _lib.ptr(planes.contiguous()), N, _lib.ptr(window_ids.contiguous()), J, W,
                                             _lib.ptr(out), 0, 0, 0, _lib.current_stream())
        _lib.check(rc, "mm_phase_diff_planes")
        return out


def phase_diff_output(phase_batch, steerable_pyramid):
    """Tester.phase_diff_output (api/tester.py:122-139): [bs, T, 13, W, H] -> (phase_0 [bs,T,24,W,H],
    phase_1 [bs,T,24,W/2,H/2])."""
    sp = steerable_pyramid
    bs, num_frames, num_phases, W, H = phase_batch.size()
    coeff_batch = sp.build_pyramid(phase_batch.reshape(bs * num_frames, num_phases, W, H))
    assert isinstance(coeff_batch, list)
    outs = []
    for c in coeff_batch:
        d = sp.extract(c)
        n, n_ch, n_ph, w, h = d.size()
        outs.append(d.view(bs, num_frames, n_ch * n_ph, w, h))
    return tuple(outs)


def phase_2_output(phase_batch, steerable_pyramid, return_phase=False):
    """The training data loaders' per-item call (Aff-wild-exps/dataloader.py:61-75, OMG-exps/dataloader.py:59-73):
    phase_batch [num_frames, num_phases, W, H] -> (phase_0 [num_frames, nbands*n_ph, W, H], phase_1 [.., W/2, H/2])."""
    sp = steerable_pyramid
    coeff_batch = sp.build_pyramid(phase_batch)
    assert isinstance(coeff_batch, list)
    outs = []
    for c in coeff_batch[:2]:
        d = sp.extract_phase(c, return_phase=return_phase)
        n, n_ch, n_ph, w, h = d.size()
        outs.append(d.view(n, -1, w, h))
    return tuple(outs)


class Steerable_Pyramid_Phase(Phase_Difference_Extractor):
    """Training-side twin of Phase_Difference_Extractor (Aff-wild-exps/utils.py:298-418, OMG-exps/utils.py): same
    constructor plus `device`, same `build_pyramid`, and `extract_phase(coeff, return_phase, return_both)`.

    The reference's own extract_phase cannot be run for a golden in the build container: its blur converts the Gaussian
    kernel to float32 only `if phase.is_cuda` (utils.py:254), so on a CPU-only host F.conv2d gets a float64 kernel and
    raises.  On CUDA it is the same arithmetic as api/utils/phase_utils.py:78-90 (float32 kernel), which IS pinned; the
    golden for this class (tests/golden/train_phase.npz) comes from running the real reference's extract_phase on the
    float64 cast of its own fp32 coefficients, where the CPU path does work."""

    def __init__(self, height=5, nbands=4, scale_factor=2, device=None, extract_level=1, visualize=False):
        Phase_Difference_Extractor.__init__(self, height, nbands, scale_factor, extract_level, visualize)
        self.device = device

    def extract_phase(self, coeff_batch, return_phase=False, return_both=False):
        """coeff [B, nbands, P, W, H, 2] -> phase differences [B, nbands, P-1, W, H] (default), the mean-centred
        denoised phase [B, nbands, P, W, H] (return_phase) or their interleave (return_both)."""
        self._check_input(coeff_batch, 6, "coeff_batch")
        B, nb, P, W, H, two = coeff_batch.shape
        assert two == 2
        if not (return_phase or return_both):
            return self.extract(coeff_batch)
        c = coeff_batch.contiguous()
        diff = torch.empty((B, nb, P - 1, W, H), dtype=torch.float32, device=c.device)
        den = torch.empty((B, nb, P, W, H), dtype=torch.float32, device=c.device)
        self._extract_generic(c, B * nb, P, W, H, diff, den)
        if return_both:
            # insert_tensors (utils.py:419-432) loops over t_a.size(dim) = P-1 positions of a 2(P-1)-long result: only its
            # first half is ever written (even slots: diff[i//2], odd slots: denoised[1 + i//2]), the rest stays zero.
            # Reproduced as is.
            L = P - 1
            both = torch.zeros((B, nb, 2 * L, W, H), dtype=torch.float32, device=c.device)
            ne, no = (L + 1) // 2, L // 2
            both[:, :, 0:L:2] = diff[:, :, :ne]
            both[:, :, 1:L:2] = den[:, :, 1:1 + no]
            return both
        return den
