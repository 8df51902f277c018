"""Fused per-video inference pipeline on one MI355X: the build's counterpart of what api/tester.py
wires together (Resnet50_Extractor.run -> Snippet_Sampler -> phase_diff_output -> Two_Stream_RNN),
without the disk round trips (.npy features, re-opened BMPs) and without the 13x redundant pyramid.

    gray frames [N,48,48] --pyramid (once per frame)--> window kernel --> phase_0 / phase_1 (NHWC)
    rgb  frames [N,3,224,224] --ResNet50 trunk--> pool5 [N,2048]
    head(phase_0, phase_1, pool5) per video (GRU over that video's snippets) --> [frames, 2]
"""
import numpy as np
import torch

from . import sampler
from .mimamo_net import Two_Stream_RNN
from .phase_difference_extractor import Phase_Difference_Extractor
from .resnet50_extractor import Resnet50_Extractor


class HotPath(object):
    def __init__(self, head_state_dict, resnet_state_dict, device=None, length=64, stride=64, num_phase=12,
                 batch_size=64):
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.length, self.stride, self.num_phase, self.batch_size = length, stride, num_phase, batch_size
        self.pde = Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
        self.resnet = Resnet50_Extractor(state_dict=resnet_state_dict, device=self.device)
        self.head = Two_Stream_RNN().load_state_dict(head_state_dict).eval().to(self.device)
        self._pre = None

    # ---- index plan for a set of videos (host, once) ---------------------------------------------
    def plan(self, video_lengths):
        """Frames of all videos are stacked along dim 0.  Returns a dict with, per video, its snippet
        ranges, and the global window-id table of every snippet frame (in snippet order)."""
        ids, vids, off = [], [], 0
        for n in video_lengths:
            ranges = sampler.snippet_ranges(n, self.length, self.stride)
            rows0 = sum(len(x) for x in ids)
            for s, e in ranges:
                ids.append(sampler.window_ids(s, e, n, self.num_phase) + off)
            vids.append({"n": n, "offset": off, "ranges": ranges, "row0": rows0,
                         "T": ranges[0][1] - ranges[0][0]})
            off += n
        ids = np.concatenate(ids, axis=0)
        # frame index (global) of every snippet row = centre column of its window
        return {"videos": vids, "ids": torch.from_numpy(ids).to(self.device).contiguous(),
                "rows": torch.from_numpy(ids[:, self.num_phase // 2].astype(np.int64)).to(self.device), "n_frames": off}

    # ---- one pass of the hot path ------------------------------------------------------------------
    def forward(self, gray, rgb, plan, independent_clips=False):
        """gray [N,48,48] f32, rgb [N,3,224,224] f32 (or NHWC4) on the device, `plan` from plan().
        Returns [rows, 2] valence/arousal for every snippet row (snippet order).

        independent_clips=True: every video is exactly one snippet of `length` frames, so each GRU call has
        seq_len 1 and the calls are batched into one (identical results: GRU batch elements are independent)."""
        J = plan["ids"].shape[0]
        if independent_clips and any(len(v["ranges"]) != 1 for v in plan["videos"]):
            raise ValueError("independent_clips=True needs single-snippet videos: a multi-snippet video's GRU runs over "
                             "its snippets (api/mimamo_net.py:119,139) and must get its own call")
        p0, cat = self.pde.phase_diff_frames(gray, plan["ids"], nhwc=True, out1_cstride=88, out1_coffset=64)
        feats = self.resnet.get_vec(rgb, channels_last4=(rgb.dim() == 4 and rgb.shape[-1] == 4))  # [N,2048], per unique frame
        rgb_rows = feats if J == feats.shape[0] and independent_clips else feats.index_select(0, plan["rows"])
        if independent_clips:
            out = self.head.forward([p0, cat], rgb_rows.view(1, J, 2048), phase_layout="nhwc_cat")
            return out.view(J, 2)
        outs = []
        for v in plan["videos"]:
            T, S = v["T"], len(v["ranges"])
            for c0 in range(0, S, self.batch_size):  # DataLoader(batch_size) chunks (api/tester.py:69-72)
                c1 = min(S, c0 + self.batch_size)
                r0, r1 = v["row0"] + c0 * T, v["row0"] + c1 * T
                o = self.head.forward([p0[r0:r1], cat[r0:r1]], rgb_rows[r0:r1].view(c1 - c0, T, 2048),
                                      phase_layout="nhwc_cat")
                outs.append(o.view(-1, 2))
        return torch.cat(outs, 0)

    # ---- lanes: independent videos on several HIP streams ------------------------------------------------
    def forward_lanes(self, inputs, video_lengths, lanes=2, independent_clips=False, from_u8=False):
        """Split the videos into `lanes` groups and run each group on its own HIP stream.  Kernels of different lanes
        interleave on the GPU, so the under-filled tail of one layer (tile quantisation: e.g. 3136 blocks on 768 slots)
        overlaps the head of another lane's layer; results are bit-identical to the single-stream pass (measured +2-3 %).
        inputs: (gray, rgb) or (frames_u8,) tensors stacked over all videos in order.  Returns [rows, 2] in video order."""
        nv = len(video_lengths)
        lanes = max(1, min(lanes, nv))
        key = (tuple(video_lengths), lanes)
        cache = getattr(self, "_lane_cache", None)
        if cache is None or cache[0] != key:
            bounds = [round(i * nv / lanes) for i in range(lanes + 1)]
            plans, frs = [], []
            off = 0
            for a, b in zip(bounds[:-1], bounds[1:]):
                n = sum(video_lengths[a:b])
                plans.append(self.plan(video_lengths[a:b]))
                frs.append((off, off + n))
                off += n
            self._lane_cache = cache = (key, plans, frs)
        _, plans, frs = cache
        # one persistent pool of side streams: per-stream workspaces (Resnet50_Extractor) are keyed by stream, so new
        # streams for every new (lengths, lanes) combination would strand tens of GB of workspace each
        pool = getattr(self, "_lane_streams", None)
        if pool is None:
            pool = self._lane_streams = []
        while len(pool) < lanes:
            pool.append(torch.cuda.Stream(device=self.device))
        streams = pool[:lanes]
        cur = torch.cuda.current_stream()
        outs = []
        for plan, (f0, f1), st in zip(plans, frs, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                part = [t[f0:f1] for t in inputs]
                o = self.forward_u8(part[0], plan, independent_clips) if from_u8 else self.forward(part[0], part[1], plan, independent_clips)
                o.record_stream(cur)
                outs.append(o)
        for st in streams:
            cur.wait_stream(st)
        return torch.cat(outs, 0)

    def forward_u8(self, frames_u8, plan, independent_clips=False):
        """Same as forward() but from the raw boundary: uint8 aligned faces [N,112,112,3] on the device
        (37.6 KB/frame over PCIe instead of 0.6 MB of fp32 tensors); PIL-exact preprocessing runs on the GPU."""
        if self._pre is None:
            from .preprocess import FramePreprocessor
            self._pre = FramePreprocessor(device=self.device)
        gray, rgb4 = self._pre(frames_u8, channels_last4=True)
        return self.forward(gray, rgb4, plan, independent_clips)

    def assemble(self, out_rows, plan, label_name=('valence', 'arousal')):
        """[rows,2] -> {video index: float64 [n_frames,2]} with the reference's overwrite order."""
        out_rows = out_rows.detach().cpu().numpy()
        res = {}
        for i, v in enumerate(plan["videos"]):
            T = v["T"]
            preds = [out_rows[v["row0"] + k * T: v["row0"] + (k + 1) * T] for k in range(len(v["ranges"]))]
            res[i] = sampler.assemble(preds, v["ranges"], len(label_name))
        return res
