"""Two_Stream_RNN on MI355X -- drop-in for api/mimamo_net.py:96-143 (inference / eval mode).

Same constructor arguments, `load_state_dict` with the reference's 107-tensor key layout, `.eval()`,
`.to(device)`, `.forward([phase_0, phase_1], rgb) -> [bs, T, 2]`.  The forward pass runs in
libmimamo_hip.so (fp32 MFMA conv engine + GRU kernels); there is no PyTorch compute path.
"""
import ctypes

import torch

from . import _lib, weights


class Two_Stream_RNN(object):
    def __init__(self, mlp_hidden_units=[2048, 256, 256], dropout=0.5, label_name='arousal_valence', num_phase=12):
        """Arguments as api/mimamo_net.py:97-122.  label_name in {'arousal', 'valence', 'arousal_valence'} sets the width
        of the output layer (len(label_name.split('_')), :120-122); mlp_hidden_units = [feature width, hidden..., 256]
        (the reference's MLP asserts the last entry, :12; here every entry must also be a multiple of 4).  num_phase sets
        PhaseNet's input channels, 2 * num_phase per level (:112); any value up to 128 is built (an odd one runs on internally
        zero-padded channel groups and takes the reference's NCHW phase layout only)."""
        num_phase = int(num_phase)
        if num_phase < 1:
            raise ValueError("num_phase must be positive")
        if num_phase > 128:
            raise NotImplementedError("num_phase must be <= 128 in this build")
        self.mlp_units = tuple(int(u) for u in mlp_hidden_units)
        assert len(self.mlp_units) - 1 > 0          # api/mimamo_net.py:11
        assert self.mlp_units[-1] == 256            # api/mimamo_net.py:12
        if any(u <= 0 or u % 4 for u in self.mlp_units):
            raise NotImplementedError("mlp_hidden_units must be positive multiples of 4 (16-byte channel groups)")
        if label_name not in ('arousal', 'valence', 'arousal_valence'):
            raise ValueError("label_name must be one of 'arousal', 'valence', 'arousal_valence' (api/mimamo_net.py:106)")
        self.label_name = label_name
        self.n_out = len(label_name.split("_"))
        self.num_phase = num_phase
        self.training = False
        self._handle = None
        self._state = None
        self._ws = {}
        self.device = None

    # -- nn.Module-like surface used by api/tester.py:45-51,77 ---------------------------------
    def state_dict(self):
        return dict(self._state) if self._state is not None else {}

    def load_state_dict(self, state_dict, strict=True):
        keys = weights.two_stream_float_keys(self.mlp_units)
        missing = [k for k in keys if k not in state_dict]
        if missing:
            raise RuntimeError("Error(s) in loading state_dict for Two_Stream_RNN: Missing key(s): %s" % missing[:4])
        if strict:
            known = set(keys) | {k + ".num_batches_tracked" for k in weights.two_stream_bn_keys(self.mlp_units)}
            extra = [k for k in state_dict if k not in known]
            if extra:
                raise RuntimeError("Error(s) in loading state_dict for Two_Stream_RNN: Unexpected key(s): %s" % extra[:4])
        for k in ("classifier.1.weight", "classifier.1.bias", "classifier.2.weight", "classifier.2.bias",
                  "classifier.2.running_mean", "classifier.2.running_var"):
            if tuple(state_dict[k].shape)[0] != self.n_out:
                raise RuntimeError("Error(s) in loading state_dict for Two_Stream_RNN: size mismatch for %s: checkpoint has %d "
                                   "outputs, label_name=%r needs %d" % (k, tuple(state_dict[k].shape)[0], self.label_name, self.n_out))
        for k, shape in weights.two_stream_shapes(self.mlp_units, self.num_phase, self.n_out).items():
            if tuple(state_dict[k].shape) != shape:     # what nn.Module.load_state_dict reports (e.g. a num_phase=12 checkpoint into num_phase=6)
                raise RuntimeError("Error(s) in loading state_dict for Two_Stream_RNN: size mismatch for %s: copying a param with "
                                   "shape %s from checkpoint, the shape in current model is %s" % (k, tuple(state_dict[k].shape), shape))
        self._state = {k: v for k, v in state_dict.items()}
        self._blob = weights.two_stream_blob(weights.widen_classifier(state_dict) if self.n_out == 1 else state_dict, self.mlp_units)
        self._release()
        return self

    def load_model_weights(self, model, model_path):  # api/mimamo_net.py:123-128
        ckp = torch.load(model_path, map_location='cpu')
        net_key = [key for key in ckp.keys() if (key != 'epoch') and (key != 'iter')][0]
        model.load_state_dict(ckp[net_key])
        return model

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is out of scope (inference hot path only)")
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError("Two_Stream_RNN (HIP) only runs on a ROCm device; there is no CPU path")
        if self.device != device:
            self._release()
        self.device = device
        return self

    def cuda(self):
        return self.to(torch.device('cuda', torch.cuda.current_device()))

    def _release(self):
        if self._handle is not None:
            _lib.lib().mm_head_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _get(self):
        if self._handle is None:
            if self._state is None:
                raise RuntimeError("load_state_dict() first: there is no default-initialised HIP model")
            if self.device is None:
                self.device = torch.device('cuda', torch.cuda.current_device())
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                units = (ctypes.c_int * len(self.mlp_units))(*self.mlp_units)
                rc = _lib.lib().mm_head_create_cfg(ctypes.byref(h), self._blob.ctypes.data_as(ctypes.c_void_p), self._blob.size,
                                                   len(self.mlp_units), units, self.num_phase)
            _lib.check(rc, "mm_head_create_cfg")
            self._handle = h
        return self._handle

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, phase_data, rgb_data, phase_layout="nchw"):
        """phase_data = [phase_0 [bs,T,C,48,48], phase_1 [bs,T,C,24,24]] with C = 2 * num_phase (24), rgb_data
        [bs,T,mlp_hidden_units[0]] -> [bs,T,n_out] (n_out = 2 for 'arousal_valence', 1 for 'arousal' / 'valence').

        phase_layout: "nchw" (reference), "nhwc" ([bs*T,48,48,C] / [bs*T,24,24,C]) or "nhwc_cat"
        (phase_1 already at channels 64..64+C of a [bs*T,24,24,64+C] buffer, completed in place).
        The GRU recurrence runs over dim 0 (bs) with T as its batch (api/mimamo_net.py:119,139)."""
        if self.training:
            raise NotImplementedError("training mode")
        h = self._get()
        phase_0, phase_1 = phase_data
        bs, T = rgb_data.size(0), rgb_data.size(1)
        for t in (phase_0, phase_1, rgb_data):
            if not t.is_cuda:
                raise RuntimeError("inputs must be on the ROCm device; this build has no CPU path")
            assert t.dtype == torch.float32
        mode = {"nchw": 0, "nhwc": 1, "nhwc_cat": 2}[phase_layout]
        C = 2 * self.num_phase
        if mode == 0:
            assert tuple(phase_0.shape) == (bs, T, C, 48, 48) and tuple(phase_1.shape) == (bs, T, C, 24, 24)
        elif mode == 1:
            assert tuple(phase_0.shape) == (bs * T, 48, 48, C) and tuple(phase_1.shape) == (bs * T, 24, 24, C)
        else:
            assert tuple(phase_0.shape) == (bs * T, 48, 48, C) and tuple(phase_1.shape) == (bs * T, 24, 24, 64 + C)
        assert rgb_data.size(2) == self.mlp_units[0]
        phase_0, phase_1, rgb = phase_0.contiguous(), phase_1.contiguous(), rgb_data.contiguous()
        out = torch.empty((bs, T, 2), dtype=torch.float32, device=rgb.device)
        L = _lib.lib()
        need = L.mm_head_workspace_bytes(h, bs, T)
        key = torch.cuda.current_stream().cuda_stream   # per-stream workspace (lanes on different streams share the handle)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            self._ws[key] = None
            ws = self._ws[key] = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=rgb.device)
        rc = L.mm_head_forward(h, _lib.ptr(phase_0), _lib.ptr(phase_1), mode, _lib.ptr(rgb), bs, T, _lib.ptr(out),
                               _lib.ptr(ws), need, _lib.current_stream())
        _lib.check(rc, "mm_head_forward")
        return out if self.n_out == 2 else out[..., :1].contiguous()

    __call__ = forward
