"""Tester on MI355X -- the build's counterpart of api/tester.py:14-139.

Same constructor keywords.  `test(video)` keeps the reference's directory contract
(`<video>_opface/<video>_aligned/frame_det_00_%06d.bmp`, api/video_processor.py:69-84) but does NOT run
OpenFace (external C++ face tracker, out of scope): the aligned-face directory must already exist -- the
reference itself skips OpenFace when it does (api/video_processor.py:64-66).  `test_frames` takes in-memory
aligned faces (what the bench uses).  Results: {video_name: DataFrame[valence, arousal]} like the reference.
"""
import os

import numpy as np
import torch

from . import sampler, synthetic, weights
from .phase_difference_extractor import phase_diff_output as _phase_diff_output
from .pipeline import HotPath


class Tester(object):
    def __init__(self, model_path, batch_size, workers=0,
                 save_size=112, nomask=True, grey=False, quiet=True, tracked_vid=False, noface_save=False,
                 OpenFace_exe='OpenFace/build/bin/FeatureExtraction',
                 benchmark_dir='pytorch-benchmarks', model_name='resnet50_ferplus_dag', feature_layer='pool5_7x7_s1',
                 num_phase=12, phase_size=48, length=64, stride=64,
                 height=4, nbands=2, scale_factor=2, extract_level=[1, 2],
                 head_state_dict=None, resnet_state_dict=None, device=None):
        """Keywords as api/tester.py:15-33.  The published configuration (num_phase 12, phase_size 48, height 4, nbands 2,
        scale_factor 2, levels [1, 2]) runs on the fused kernels; other sampler / pyramid configurations are forwarded to
        the general pyramid + generic extract kernels in the reference's windowed form.  Like the reference, the model is
        always Two_Stream_RNN() (api/tester.py:44): a configuration only runs if it hands PhaseNet 2 x 12 channels at 48 x 48
        and 24 x 24 (e.g. nbands=4, num_phase=6); anything else fails at the model's input check, as it does there."""
        self.batch_size, self.workers, self.save_size = batch_size, workers, save_size
        self.num_phase, self.phase_size, self.length, self.stride = num_phase, phase_size, length, stride
        self.label_name = ['valence', 'arousal']  # api/tester.py:52
        if head_state_dict is None:
            assert os.path.exists(model_path)  # api/tester.py:46
            checkpoint = torch.load(model_path, map_location='cpu')
            head_state_dict = checkpoint['state_dict']
            print("load checkpoint from {}, epoch:{}".format(model_path, checkpoint['epoch']))
        if resnet_state_dict is None:
            pth = os.path.join(os.path.abspath(benchmark_dir), 'ferplus', model_name + '.pth')
            assert os.path.exists(pth), 'benchmark_dir must exits'
            resnet_state_dict = torch.load(pth, map_location='cpu')
        self.hot = HotPath(head_state_dict, resnet_state_dict, device, length, stride, num_phase, batch_size, phase_size=phase_size,
                           height=height, nbands=nbands, scale_factor=scale_factor, extract_level=tuple(extract_level),
                           resnet_kwargs=dict(benchmark_dir=benchmark_dir, model_name=model_name, feature_layer=feature_layer))
        self.device = self.hot.device
        self.phase_difference_extractor = self.hot.pde
        self.resnet50_extractor = self.hot.resnet
        self.model = self.hot.head

    # -- reference surface ---------------------------------------------------------------------------
    def phase_diff_output(self, phase_batch, steerable_pyramid):
        return _phase_diff_output(phase_batch, steerable_pyramid)

    def test_on_dataloader(self, dataloader, model=None, train_mean=None, train_std=None):
        """Reference loop (api/tester.py:76-121) over batches `(phase_f [bs,T,13,48,48], rgb_f [bs,T,2048], label,
        ranges [bs,2], names [bs])` as Snippet_Sampler + DataLoader produce them: windowed phase input (the 13x
        redundant form), one GRU call per batch, assembly with later snippets overwriting earlier ones."""
        import pandas as pd
        if train_mean is not None or train_std is not None:
            raise NotImplementedError("train_mean/train_std rescaling calls an undefined `correct` in the reference (tester.py:100)")
        model = self.model if model is None else model
        model.eval()
        sample_names, sample_preds, sample_ranges = [], [], []
        for data_batch in dataloader:
            phase_f, rgb_f, _, ranges, names = data_batch
            with torch.no_grad():
                phase_f = torch.as_tensor(phase_f).float().to(self.device)
                phase_0, phase_1 = self.phase_diff_output(phase_f, self.phase_difference_extractor)
                rgb_f = torch.as_tensor(rgb_f).float().to(self.device)
                output = model([phase_0, phase_1], rgb_f)
            sample_names.append(np.asarray(names))
            sample_ranges.append(np.asarray(ranges))
            sample_preds.append(output.cpu().numpy())
        sample_names = np.concatenate(sample_names, axis=0)
        sample_preds = np.concatenate(sample_preds, axis=0)
        sample_ranges = np.concatenate(sample_ranges, axis=0)
        n_labels = sample_preds.shape[-1]
        video_dict = {}
        for video in sample_names:
            if video in video_dict:
                continue
            mask = sample_names == video
            video_ranges, video_preds = sample_ranges[mask], sample_preds[mask]
            max_len = max(r[-1] for r in video_ranges)
            arr = np.zeros((max_len, n_labels))
            min_f, max_f = 0, 0
            for (start, end), pred in zip(video_ranges, video_preds):
                arr[start:end, :] = pred
                min_f, max_f = min(min_f, start), max(max_f, end)
            assert (min_f == 0) and (max_f == max_len)
            video_dict[video] = pd.DataFrame(data=arr, columns=self.label_name)
        return video_dict

    def test(self, input_video):
        import pandas as pd
        video_name = os.path.basename(input_video).split('.')[0]
        opface_output_dir = os.path.join(os.path.dirname(input_video), video_name + "_opface")
        if not os.path.isdir(os.path.join(opface_output_dir, video_name + "_aligned")):
            raise RuntimeError("aligned faces not found under %s: run OpenFace FeatureExtraction first "
                               "(api/video_processor.py:69-84); the face tracker is outside this build" % opface_output_dir)
        frames = sampler.list_aligned_frames(opface_output_dir, video_name)
        paths = [p for _, p in frames]
        u8 = sampler.load_u8_batch(paths, self.save_size)
        if u8 is not None:
            # OpenFace's -simsize 112 crops (api/video_processor.py:75): decode only on the host, resize / crop /
            # normalise on the GPU -- bit-exact with the PIL calls of the reference's samplers (csrc/preproc.hip)
            return self.test_frames([u8], names=[video_name])
        gray = sampler.load_gray_batch(paths, self.phase_size).to(self.device)   # other frame sizes: PIL on the host
        rgb = sampler.load_rgb_batch(paths).to(self.device)
        res = self._run([len(paths)], gray, rgb)
        return {video_name: pd.DataFrame(data=res[0], columns=self.label_name)}

    def test_frames(self, clips_u8, names=None):
        """clips_u8: list of uint8 arrays [n_i,112,112,3] (aligned faces).  -> {name: DataFrame}.
        The frames stay on the host (pinned) and are streamed to the GPU chunk by chunk under the compute of the previous
        chunk (HotPath._forward_u8_host): a long video never sits on the device as a whole."""
        import pandas as pd
        from .stream import pin
        frames = pin(np.concatenate(clips_u8) if len(clips_u8) != 1 else np.asarray(clips_u8[0]))
        plan = self.hot.plan([len(c) for c in clips_u8])
        with torch.no_grad():
            out = self.hot.forward_u8(frames, plan)  # upload + PIL-exact preprocessing on the GPU
        res = self.hot.assemble(out, plan, self.label_name)
        names = names or ["clip%d" % i for i in range(len(clips_u8))]
        return {names[i]: pd.DataFrame(data=res[i], columns=self.label_name) for i in range(len(clips_u8))}

    def _run(self, lengths, gray, rgb):
        plan = self.hot.plan(lengths)
        with torch.no_grad():
            out = self.hot.forward(gray, rgb, plan)
        return self.hot.assemble(out, plan, self.label_name)
