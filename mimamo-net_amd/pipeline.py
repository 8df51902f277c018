"""Fused per-video inference pipeline on one MI355X: the build's counterpart of what api/tester.py
wires together (Resnet50_Extractor.run -> Snippet_Sampler -> phase_diff_output -> Two_Stream_RNN),
without the disk round trips (.npy features, re-opened BMPs) and without the 13x redundant pyramid.

    gray frames [N,48,48] --pyramid (once per frame)--> window kernel --> phase_0 / phase_1 (NHWC)
    rgb  frames [N,3,224,224] --ResNet50 trunk--> pool5 [N,2048]
    head(phase_0, phase_1, pool5) per video (GRU over that video's snippets) --> [frames, 2]

Memory is bounded by the chunk sizes, not by the video length: the reference streams 64 frames at a time
(api/resnet50_extractor.py:56-60, api/tester.py:69-72); here preprocessing + ResNet50 run over at most
`max_frames_per_call` frames per call (0.8 MB of NHWC4 input + 17.3 MB of activations per frame) and the phase
stage + head over row groups of at most `batch_size` snippets, so an hour-long video needs the same workspaces as
a one-minute one plus 8 KB + 9 KB per frame for its features and gray frames.
"""
import numpy as np
import torch

from . import sampler
from .mimamo_net import Two_Stream_RNN
from .phase_difference_extractor import Phase_Difference_Extractor
from .resnet50_extractor import Resnet50_Extractor


class HotPath(object):
    PUBLISHED = (12, 48, 4, 2, 2, (1, 2))      # num_phase, phase_size, height, nbands, scale_factor, extract_level (api/tester.py:28-32)

    def __init__(self, head_state_dict, resnet_state_dict, device=None, length=64, stride=64, num_phase=12,
                 batch_size=64, max_frames_per_call=4096, upload_chunk_frames=1024, phase_size=48, height=4, nbands=2,
                 scale_factor=2, extract_level=(1, 2), model_num_phase=12, resnet_kwargs=None):
        """resnet_kwargs: further Resnet50_Extractor keywords (benchmark_dir / model_name: where a model definition file may sit;
        stride_on_first_1x1, ceil_mode, bn_eps, mean).
        model_num_phase: num_phase of the Two_Stream_RNN the checkpoint belongs to.  The reference's Tester always builds
        Two_Stream_RNN() with its default (api/tester.py:44), whatever num_phase its sampler uses -- so a non-published
        sampler / pyramid configuration only runs there (and here) when it still hands PhaseNet 24 channels at 48x48 and
        24x24, e.g. nbands=4 with num_phase=6."""
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.length, self.stride, self.num_phase, self.batch_size = length, stride, num_phase, batch_size
        self.max_frames_per_call = int(max_frames_per_call)
        self.phase_size = int(phase_size)
        levels = tuple(extract_level) if not isinstance(extract_level, int) else (extract_level,)
        # the fused kernels (one pyramid per unique frame) implement the published configuration; anything else runs the
        # reference's windowed form on the general pyramid + generic extract kernels (13x redundant like the reference, not tuned)
        self.fused = (int(num_phase), self.phase_size, int(height), int(nbands), int(scale_factor), levels) == self.PUBLISHED
        self.upload_chunk_frames = max(1, int(upload_chunk_frames))   # host-resident input: frames per double-buffered upload
        self._feeders = {}
        self.pde = Phase_Difference_Extractor(height, nbands, scale_factor, list(levels) if len(levels) > 1 else levels[0], False)
        self.resnet = Resnet50_Extractor(state_dict=resnet_state_dict, device=self.device,
                                         max_frames_per_call=self.max_frames_per_call, **(resnet_kwargs or {}))
        self.head = Two_Stream_RNN(num_phase=model_num_phase).load_state_dict(head_state_dict).eval().to(self.device)
        self._pre = None

    # ---- index plan for a set of videos (host, once) ---------------------------------------------
    def plan(self, video_lengths):
        """Frames of all videos are stacked along dim 0.  Returns a dict with, per video, its snippet ranges and
        the position of its rows in the [rows, 2] result (snippet order, as the reference's sampler emits them),
        and the row GROUPS the phase stage + head are run over.

        A group is one head call: the GRU's recurrence runs over the snippets of ONE video's DataLoader batch
        (api/mimamo_net.py:119,139; api/tester.py:69-72), its batch dimension over the frames of a snippet.  GRU batch
        elements are independent, so consecutive videos with the same snippet count S (<= batch_size) and snippet length
        T share a call: their rows are laid out [s][video][t] and the call sees bs = S, T' = n_videos * T -- the same
        bits as one call per video.  Independent 64-frame clips are the S = 1 case.  A video with more than batch_size
        snippets is split into the reference's batches of batch_size snippets, each its own call."""
        vids, groups = [], []
        off, row0 = 0, 0
        for n in video_lengths:
            ranges = sampler.snippet_ranges(n, self.length, self.stride)
            T = ranges[0][1] - ranges[0][0]
            vids.append({"n": n, "offset": off, "ranges": ranges, "row0": row0, "T": T})
            off += n
            row0 += len(ranges) * T
        cap_rows = max(self.batch_size * self.length, 1)
        i = 0
        while i < len(vids):
            v = vids[i]
            S, T = len(v["ranges"]), v["T"]
            if S > self.batch_size:
                for c0 in range(0, S, self.batch_size):
                    c1 = min(S, c0 + self.batch_size)
                    groups.append(self._group([v], c0, c1))
                i += 1
                continue
            j, rows = i + 1, S * T
            while (j < len(vids) and len(vids[j]["ranges"]) == S and vids[j]["T"] == T and rows + S * T <= cap_rows):
                rows += S * T
                j += 1
            groups.append(self._group(vids[i:j], 0, S))
            i = j
        return {"videos": vids, "groups": groups, "n_frames": off, "n_rows": row0}

    def _group(self, vs, c0, c1):
        """Rows of snippets [c0, c1) of the videos `vs` in [s][video][t] order."""
        T = vs[0]["T"]
        ids, dest = [], []
        for s in range(c0, c1):
            for v in vs:
                a, b = v["ranges"][s]
                ids.append(sampler.window_ids(a, b, v["n"], self.num_phase) + v["offset"])
                dest.append(np.arange(v["row0"] + s * T, v["row0"] + (s + 1) * T, dtype=np.int64))
        ids = np.concatenate(ids, axis=0)
        dest = np.concatenate(dest)
        f0, f1 = int(ids.min()), int(ids.max()) + 1
        rows = ids[:, self.num_phase // 2].astype(np.int64)      # frame of a row = centre column of its window
        contiguous = bool((dest == np.arange(dest[0], dest[0] + len(dest))).all())
        return {"bs": c1 - c0, "T": len(vs) * T, "f0": f0, "f1": f1,
                "ids": torch.from_numpy(ids - f0).to(self.device).contiguous(),
                "rows": None if (contiguous and bool((rows == np.arange(rows[0], rows[0] + len(rows))).all()))
                else torch.from_numpy(rows).to(self.device),
                "row_first": int(rows[0]),
                "dest": None if contiguous else torch.from_numpy(dest).to(self.device),
                "dest_first": int(dest[0]), "n": len(dest)}

    # ---- one pass of the hot path ------------------------------------------------------------------
    def forward(self, gray, rgb, plan, independent_clips=False):
        """gray [N,48,48] f32, rgb [N,3,224,224] f32 (or NHWC4) on the device, `plan` from plan().
        Returns [rows, 2] valence/arousal for every snippet row (snippet order).

        independent_clips=True asserts that every video is exactly one snippet (the bench's workload); the batching of
        their GRU calls is what plan() does for any run of equal-shaped videos."""
        self._check(plan, gray.shape[0], independent_clips)
        feats = self.resnet.get_vec(rgb, channels_last4=(rgb.dim() == 4 and rgb.shape[-1] == 4))  # [N,2048], per unique frame
        return self._rows(gray, feats, plan)

    def forward_u8(self, frames_u8, plan, independent_clips=False):
        """Same as forward() but from the raw boundary: uint8 aligned faces [N,112,112,3] on the device
        (37.6 KB/frame over PCIe instead of 0.6 MB of fp32 tensors); PIL-exact preprocessing runs on the GPU,
        chunk by chunk in front of the ResNet50 trunk so the fp32 RGB tensor never exists for more than one chunk."""
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or tuple(frames_u8.shape[1:]) != (112, 112, 3):
            raise ValueError("frames must be uint8 [N,112,112,3]")
        self._check(plan, frames_u8.shape[0], independent_clips)
        if self._pre is None:
            from .preprocess import FramePreprocessor
            self._pre = FramePreprocessor(phase_size=self.phase_size, mean=self.resnet.meta['mean'], device=self.device)
        if not frames_u8.is_cuda:
            if not frames_u8.is_pinned():
                # a pageable source makes every non_blocking copy synchronous with the host: nothing would overlap
                from .stream import pin
                frames_u8 = pin(frames_u8)
            return self._forward_u8_host(frames_u8, plan)
        N, step = frames_u8.shape[0], self.max_frames_per_call
        if N <= step:
            gray, rgb3 = self._pre(frames_u8, bordered3=True)
            feats = self.resnet.get_vec(rgb3)
        else:
            gray = torch.empty((N, self._pre.phase_size, self._pre.phase_size), dtype=torch.float32, device=frames_u8.device)
            feats = torch.empty((N, 2048), dtype=torch.float32, device=frames_u8.device)
            for c0 in range(0, N, step):
                c1 = min(N, c0 + step)
                g, rgb3 = self._pre(frames_u8[c0:c1], bordered3=True)
                gray[c0:c1] = g
                self.resnet.get_vec(rgb3, out=feats[c0:c1])
        return self._rows(gray, feats, plan)

    def _forward_u8_host(self, frames, plan):
        """frames_u8 on the HOST (pinned: stream.pin): uploaded `upload_chunk_frames` at a time on a copy stream, chunk c+1 in
        flight while preprocessing + ResNet50 run on chunk c (stream.FrameStream) -- the build's counterpart of the reference's
        per-batch DataLoader feed (api/resnet50_extractor.py:53-72, api/tester.py:65-73).  Same rows as the device-resident
        call, bit for bit (frames are independent through preprocessing and the trunk)."""
        from .stream import FrameStream
        N = frames.shape[0]
        chunk = min(self.max_frames_per_call, self.upload_chunk_frames)
        key = torch.cuda.current_stream().cuda_stream
        fs = self._feeders.get(key)
        if fs is None or fs.slots[0].shape[0] != chunk:
            if fs is not None:
                fs.close()
            if len(self._feeders) > 8:
                for old in self._feeders.values():
                    old.close()
                self._feeders.clear()
            fs = self._feeders[key] = FrameStream(self.device, chunk)
        gray = torch.empty((N, self._pre.phase_size, self._pre.phase_size), dtype=torch.float32, device=self.device)
        feats = torch.empty((N, 2048), dtype=torch.float32, device=self.device)
        starts = list(range(0, N, chunk))
        fs.upload(0, [frames[0:min(N, chunk)]])
        for i, c0 in enumerate(starts):
            c1, slot = min(N, c0 + chunk), i % 2
            if i + 1 < len(starts):
                n0 = starts[i + 1]
                fs.upload((i + 1) % 2, [frames[n0:min(N, n0 + chunk)]])
            dev = fs.acquire(slot)
            g, rgb3 = self._pre(dev, bordered3=True)
            gray[c0:c1] = g
            self.resnet.get_vec(rgb3, out=feats[c0:c1])
            fs.release(slot)
        return self._rows(gray, feats, plan)

    def _check(self, plan, n_frames, independent_clips):
        if n_frames != plan["n_frames"]:
            raise ValueError("plan was built for %d frames, got %d" % (plan["n_frames"], n_frames))
        if independent_clips and any(len(v["ranges"]) != 1 for v in plan["videos"]):
            raise ValueError("independent_clips=True needs single-snippet videos: a multi-snippet video's GRU runs over "
                             "its snippets (api/mimamo_net.py:119,139)")

    def _rows(self, gray, feats, plan):
        groups = plan["groups"]
        out = None
        for g in groups:
            rgb_rows = (feats[g["row_first"]:g["row_first"] + g["n"]] if g["rows"] is None
                        else feats.index_select(0, g["rows"]))
            if self.fused:
                p0, cat = self.pde.phase_diff_frames(gray[g["f0"]:g["f1"]], g["ids"], nhwc=True, out1_cstride=88, out1_coffset=64,
                                                     ids_checked=True)
                o = self.head.forward([p0, cat], rgb_rows.view(g["bs"], g["T"], 2048), phase_layout="nhwc_cat").view(-1, 2)
            else:
                # the reference's own form (api/tester.py:122-139): windows gathered per row, pyramid per window frame
                # The gathered windows are (num_phase + 1) x the frames: run the pyramid over at most max_frames_per_call window
                # frames at a time (rows are independent through phase_diff_output), so memory follows the chunk, not the video
                from .phase_difference_extractor import phase_diff_output
                P = self.num_phase + 1
                rows_per = max(1, self.max_frames_per_call // P)
                src = gray[g["f0"]:g["f1"]]
                parts = None
                for r0 in range(0, g["n"], rows_per):
                    r1 = min(g["n"], r0 + rows_per)
                    win = src.index_select(0, g["ids"][r0:r1].reshape(-1).long()).view(1, r1 - r0, P, self.phase_size, self.phase_size)
                    lv = phase_diff_output(win, self.pde)
                    if len(lv) != 2:
                        raise ValueError("Two_Stream_RNN takes two pyramid levels (api/mimamo_net.py:133); extract_level gave %d" % len(lv))
                    if parts is None:
                        parts = [torch.empty((g["n"],) + tuple(l.shape[2:]), dtype=l.dtype, device=l.device) for l in lv]
                    for dst, l in zip(parts, lv):
                        dst[r0:r1] = l[0]
                levels = [p.view((g["bs"], g["T"]) + tuple(p.shape[1:])) for p in parts]
                o = self.head.forward(levels, rgb_rows.view(g["bs"], g["T"], 2048)).view(-1, 2)
            if len(groups) == 1 and g["dest"] is None:
                return o
            if out is None:
                out = torch.empty((plan["n_rows"], 2), dtype=torch.float32, device=feats.device)
            if g["dest"] is None:
                out[g["dest_first"]:g["dest_first"] + g["n"]] = o
            else:
                out.index_copy_(0, g["dest"], o)
        return out

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

    def assemble(self, out_rows, plan, label_name=('valence', 'arousal')):
        """[rows,2] -> {video index: float64 [n_frames,2]} with the reference's overwrite order."""
        out_rows = out_rows.detach().cpu().numpy()
        res = {}
        for i, v in enumerate(plan["videos"]):
            T = v["T"]
            preds = [out_rows[v["row0"] + k * T: v["row0"] + (k + 1) * T] for k in range(len(v["ranges"]))]
            res[i] = sampler.assemble(preds, v["ranges"], len(label_name))
        return res
