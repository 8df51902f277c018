"""Resnet50_Extractor on MI355X -- drop-in for api/resnet50_extractor.py:13-88.

The reference loads a third-party model definition + weights by name
(`pytorch-benchmarks/ferplus/resnet50_ferplus_dag.{py,pth}`, api/utils/model_utils.py:65-79) and hooks
layer `pool5_7x7_s1`.  That dependency is not vendored; here the same Caffe-style ResNet-50 trunk runs on
the fp32 MFMA conv engine of libmimamo_hip.so, with weights taken from a state_dict in the third-party
file's key layout (`conv1_7x7_s2.weight`, `conv1_7x7_s2_bn.{weight,bias,running_mean,running_var}`, ...).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, weights


class Resnet50_Extractor(object):
    def __init__(self, benchmark_dir='pytorch-benchmarks', model_name='resnet50_ferplus_dag',
                 feature_layer='pool5_7x7_s1', state_dict=None, device=None, max_frames_per_call=4096,
                 stride_on_first_1x1=None, ceil_mode=None, bn_eps=None, mean=None):
        """benchmark_dir/model_name/feature_layer as api/resnet50_extractor.py:14-15.

        stride_on_first_1x1 / ceil_mode / bn_eps / mean: the parts of the third-party graph the reference's own code does not
        pin: stride 2 of a stage's first block on its 1x1 reduce conv (Caffe style, True) or on its 3x3 conv
        (False); `pool1_3x3_s2` with ceil_mode (112 -> 56) or floor (-> 55); BatchNorm eps; `meta['mean']`.  None = what the model
        definition file `<benchmark_dir>/ferplus/<model_name>.py` says when it exists (the file the reference executes,
        api/utils/model_utils.py:65-79; read with `ast`, see weights.read_model_definition), else the published model's values
        (True, True, 1e-5, weights.RESNET50_MEAN).

        state_dict: weights in the third-party key layout.  If None, `<benchmark_dir>/ferplus/<model_name>.pth`
        is loaded when it exists (the reference's location, api/resnet50_extractor.py:35-36); otherwise the
        constructor asserts like the reference does (:33).
        max_frames_per_call: get_vec runs the trunk over at most this many frames per launch sequence (17.3 MB of
        activation workspace per frame), so memory does not grow with the batch the caller hands over; the reference
        streams 64 frames at a time (:56-60)."""
        if feature_layer != 'pool5_7x7_s1':
            raise NotImplementedError("only the pool5_7x7_s1 hook of the reference is implemented")
        self.benchmark_dir = os.path.abspath(benchmark_dir)
        self.model_name = model_name
        self.feature_layer = feature_layer
        self.max_frames_per_call = max(1, int(max_frames_per_call))
        if state_dict is None:
            assert os.path.exists(self.benchmark_dir), 'benchmark_dir must exits'
            pth = os.path.join(self.benchmark_dir, 'ferplus', model_name + '.pth')
            state_dict = torch.load(pth, map_location='cpu')
        definition = {}
        def_path = os.path.join(self.benchmark_dir, 'ferplus', model_name + '.py')
        if os.path.isfile(def_path):
            definition = weights.read_model_definition(def_path)
        pick = lambda given, key, default: given if given is not None else definition.get(key, default)   # noqa: E731
        self.stride_on_first_1x1 = bool(pick(stride_on_first_1x1, 'stride_on_first_1x1', True))
        self.ceil_mode = bool(pick(ceil_mode, 'ceil_mode', True))
        self.bn_eps = float(pick(bn_eps, 'bn_eps', 1e-5))
        self.meta = {'mean': list(weights.RESNET50_MEAN), 'std': [1, 1, 1], 'imageSize': [224, 224, 3]}
        self.meta.update(definition.get('meta', {}))
        if mean is not None:
            self.meta['mean'] = [float(m) for m in mean]
        if list(self.meta['std']) != [1, 1, 1] or list(self.meta['imageSize'])[:2] != [224, 224] or len(self.meta['mean']) != 3:
            # compose_transforms (api/utils/model_utils.py:26-39) would scale / crop differently; the GPU preprocessing is built for this one
            raise NotImplementedError("model meta %r: only std [1,1,1] and imageSize 224 are implemented" % (self.meta,))
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        blob = weights.resnet50_blob(state_dict)
        h = ctypes.c_void_p()
        L = _lib.lib()
        with torch.cuda.device(self.device):
            rc = L.mm_resnet50_create(ctypes.byref(h), blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                      int(self.stride_on_first_1x1), int(self.ceil_mode), self.bn_eps)
        _lib.check(rc, "mm_resnet50_create")
        self._handle = h
        self._ws = {}   # per-stream workspaces: the handle itself is stateless, so lanes on different streams may share it
        self._pre = None  # FramePreprocessor of run(), created on first use

    def set_winograd(self, mode=True):
        """Algorithm of the stride-1 3x3 layers of conv2_x..conv5_x: True/1 = default (Winograd F(4x4,3x3): variant 5 for
        conv2_x..conv4_x, variant 4 for conv5_x),
        2 = F(2x2,3x3), 4 = F(4x4,3x3) as three kernels (input transform, 36 batched GEMMs, output transform),
        5 = F(4x4,3x3) with the output transform fused into the position GEMMs (csrc/wino_fused.hip),
        False/0 = direct implicit GEMM."""
        mode = {True: 1, False: 0}.get(mode, mode)
        _lib.check(_lib.lib().mm_resnet50_set_winograd(self._handle, int(mode)), "mm_resnet50_set_winograd")

    def set_precision(self, mode="fp32"):
        """"fp32" (default; the reference's arithmetic, the headline and every parity claim) or "bf16x3": the 1x1 layers with K >= 512
        on the bf16 matrix pipes through a three-way bf16 split of both fp32 operands (mm_resnet50_set_precision; bench.py extra.bf16x3)."""
        _lib.check(_lib.lib().mm_resnet50_set_precision(self._handle, {"fp32": 0, "bf16x3": 1}[mode]), "mm_resnet50_set_precision")

    def close(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().mm_resnet50_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _workspace(self, bs):
        need = _lib.lib().mm_resnet50_workspace_bytes(self._handle, bs)
        key = torch.cuda.current_stream().cuda_stream
        ws = self._ws.pop(key, None)          # re-inserted below: the dict is kept in least-recently-used order
        if ws is None or ws.numel() * 4 < need:
            ws = None
            while len(self._ws) >= 8:         # bound what short-lived streams can strand (17 MB per frame each)
                self._ws.pop(next(iter(self._ws)))
            ws = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=self.device)
        self._ws[key] = ws
        return ws, need

    def get_vec(self, image, channels_last4=False, out=None):
        """image [bs,3,224,224] (255*x - mean) -> pool5 features [bs,2048] ON THE DEVICE.

        (api/resnet50_extractor.py:74-83 returns relu(squeeze()) of a CPU copy; squeeze() collapsing bs=1 and
        the host copy are quirk Q8 and are not reproduced.)  channels_last4=True takes [bs,224,224,4]; a [bs,230,230,3] tensor is
        the zero-bordered packed layout FramePreprocessor(bordered3=True) writes.
        out: optional preallocated [bs,2048] fp32 device tensor (contiguous rows) the features are written to."""
        if not image.is_cuda:
            raise RuntimeError("image must be on the ROCm device; this build has no CPU path")
        assert image.dtype == torch.float32
        bs = image.size(0)
        if tuple(image.shape[1:]) == (230, 230, 3):
            mode = 2      # zero-bordered packed NHWC3 (FramePreprocessor(bordered3=True)): stem with K = 168
        else:
            assert tuple(image.shape[1:]) == ((224, 224, 4) if channels_last4 else (3, 224, 224))
            mode = 0 if channels_last4 else 1
        image = image.contiguous()
        if out is None:
            out = torch.empty((bs, 2048), dtype=torch.float32, device=image.device)
        else:
            assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (bs, 2048) and out.is_contiguous()
        step = self.max_frames_per_call
        ws, need = self._workspace(min(bs, step))
        for c0 in range(0, bs, step):
            c1 = min(bs, c0 + step)
            rc = _lib.lib().mm_resnet50_forward(self._handle, _lib.ptr(image[c0:c1]), mode, c1 - c0,
                                                _lib.ptr(out[c0:c1]), _lib.ptr(ws), need, _lib.current_stream())
            _lib.check(rc, "mm_resnet50_forward")
        return out

    def run(self, input_dir, output_dir, batch_size=64, video_name=''):
        """Write one `%05d.npy` (f32[2048]) per aligned face `frame_det_00_%06d.bmp` (api/resnet50_extractor.py:42-73);
        skipped when output_dir already holds .npy files (:61-65)."""
        from .sampler import list_aligned_frames, load_rgb_batch
        assert os.path.exists(input_dir), 'input dir must exsit!'
        assert len(os.listdir(input_dir)) != 0, 'input dir must not be empty!'
        assert len(video_name) != 0, 'input video name cannot be empty!'
        if not os.path.exists(output_dir):
            os.makedirs(output_dir)
        elif len(os.listdir(output_dir)) != 0 and '.npy' in os.listdir(output_dir)[0]:
            print("output_dir {} already exists, feature extraction skipped.".format(output_dir))
            return
        from .sampler import load_u8_batch
        from .stream import pin
        frames = list_aligned_frames(input_dir, video_name)
        for i in range(0, len(frames), batch_size):
            chunk = frames[i:i + batch_size]
            paths = [p for _, p in chunk]
            u8 = load_u8_batch(paths, 112)
            if u8 is not None:
                # OpenFace's -simsize 112 crops (api/video_processor.py:75): decode on the host, resize / crop / normalise on the GPU
                # -- bit-exact with the PIL calls of utils/model_utils.py:29-39 (csrc/preproc.hip), 37.6 KB instead of 602 KB
                # per frame over PCIe
                if self._pre is None:
                    from .preprocess import FramePreprocessor
                    self._pre = FramePreprocessor(mean=self.meta['mean'], device=self.device)
                _, rgb3 = self._pre(pin(u8).to(self.device, non_blocking=True), want_gray=False, bordered3=True)
                feats = self.get_vec(rgb3).cpu().numpy()
            else:                                                  # other frame sizes: PIL on the host, as the reference does
                ims = load_rgb_batch(paths, self.meta['mean']).to(self.device)
                feats = self.get_vec(ims).cpu().numpy()
            for (idx, _), f in zip(chunk, feats):
                np.save(os.path.join(output_dir, "%05d.npy" % idx), f)

    @staticmethod
    def get_frame_index(frame_path):
        return int(os.path.basename(frame_path).split('.')[0].split('_')[-1])
