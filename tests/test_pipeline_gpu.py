"""GPU parity of the whole per-video path (Tester / HotPath) against the oracle driven with the reference's
semantics: windowed 13x-redundant pyramid, ResNet50 features per frame, one GRU call per video with its snippets
in sampler order (recurrence over snippets), tail-snippet overwrite."""
import os

import numpy as np
import pytest
import torch

from mimamo_net_amd import synthetic, weights, sampler

pytestmark = pytest.mark.gpu
OUT_ATOL = 1e-4


@pytest.fixture(scope="module")
def tester(pkg):
    from mimamo_net_amd.tester import Tester
    return Tester(model_path=None, batch_size=64, head_state_dict=weights.make_two_stream_state_dict(seed=0),
                  resnet_state_dict=weights.make_resnet50_state_dict(seed=0), device="cuda:0")


def _oracle_video(oracle, clip_u8, length=64, stride=64, batch_size=64):
    gray, rgb = synthetic.preprocess_host(clip_u8)
    n = len(clip_u8)
    ranges = oracle.snippet_ranges(n, length, stride)
    feats = oracle.resnet50_pool5(weights.make_resnet50_state_dict(seed=0), rgb)
    sd = weights.make_two_stream_state_dict(seed=0)
    T = ranges[0][1] - ranges[0][0]
    ph = np.stack([gray[oracle.window_ids(s, e, n)] for s, e in ranges])      # [S,T,13,48,48]
    rg = np.stack([feats[s:e] for s, e in ranges])                             # [S,T,2048]
    preds = []
    for c0 in range(0, len(ranges), batch_size):
        p0, p1 = oracle.phase_diff_output(ph[c0:c0 + batch_size])
        preds.extend(list(oracle.two_stream_forward(sd, p0, p1, rg[c0:c0 + batch_size])))
    return oracle.assemble(preds, ranges)


def test_multi_snippet_video_and_short_video(tester, oracle):
    """150 frames -> snippets [0,64) [64,128) [86,150): GRU seq_len 3 + overwrite order; 20 frames -> short-video rule."""
    clips = [synthetic.make_clip_u8(40, 150), synthetic.make_clip_u8(41, 20)]
    res = tester.test_frames(clips, names=["a", "b"])
    assert list(res["a"].columns) == ["valence", "arousal"] and res["a"].shape == (150, 2) and res["b"].shape == (20, 2)
    for name, clip in zip(("a", "b"), clips):
        want = _oracle_video(oracle, clip)
        err = np.abs(res[name].values - want).max()
        assert err < OUT_ATOL, (name, err)


def test_overlapping_snippets_and_tiny_videos(oracle):
    """Tester(length=32, stride=16): overlapping snippets -- GRU seq_len 6 for a 100-frame video, later snippets overwrite the frames
    they share with earlier ones (api/tester.py:113-116) -- and videos of one and two frames (every window clamped to the video:
    exactly zero phase differences), against the oracle driven with the same keywords."""
    from mimamo_net_amd.tester import Tester
    head_sd, resnet_sd = weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0)
    t = Tester(model_path=None, batch_size=64, head_state_dict=head_sd, resnet_state_dict=resnet_sd, device="cuda:0", length=32, stride=16)
    clips = [synthetic.make_clip_u8(90, 100), synthetic.make_clip_u8(91, 1), synthetic.make_clip_u8(92, 2)]
    res = t.test_frames(clips, names=["a", "one", "two"])
    for name, clip in zip(("a", "one", "two"), clips):
        want = _oracle_video(oracle, clip, length=32, stride=16)
        err = np.abs(res[name].values - want).max()
        print("length 32 / stride 16, %d frames: %.2e" % (len(clip), err))
        assert res[name].shape == (len(clip), 2) and err < OUT_ATOL, (name, err)


def test_independent_clip_batching_is_exact(tester):
    """Batching single-snippet clips into one GRU call (seq_len 1) gives the same bits as one call per clip."""
    clips = [synthetic.make_clip_u8(50 + i, 64) for i in range(3)]
    frames = torch.from_numpy(np.concatenate(clips)).to(tester.device)
    plan = tester.hot.plan([64, 64, 64])
    with torch.no_grad():
        a = tester.hot.forward_u8(frames, plan, independent_clips=True)
        b = tester.hot.forward_u8(frames, plan, independent_clips=False)
    assert torch.equal(a, b)


def test_tester_reads_reference_directory_layout(tester, oracle, tmp_path):
    """<video>_opface/<video>_aligned/frame_det_00_%06d.bmp (api/video_processor.py:69-84) + %05d.npy features."""
    from PIL import Image
    clip = synthetic.make_clip_u8(60, 12)
    vid = tmp_path / "utt.mp4"
    al = tmp_path / "utt_opface" / "utt_aligned"
    os.makedirs(al)
    for i, f in enumerate(clip):
        Image.fromarray(f, "RGB").save(str(al / ("frame_det_00_%06d.bmp" % (i + 1))))
    res = tester.test(str(vid))
    assert list(res) == ["utt"] and res["utt"].shape == (12, 2)
    want = _oracle_video(oracle, clip)
    assert np.abs(res["utt"].values - want).max() < OUT_ATOL
    # test() decodes on the host and preprocesses on the GPU; the host-side PIL preprocessing (used for frame sizes
    # other than save_size) gives the same bits, because the GPU kernels are bit-exact with PIL and the NCHW entry point
    # converts to the same packed-row layout the uint8 path writes
    paths = [p for _, p in sampler.list_aligned_frames(str(tmp_path / "utt_opface"), "utt")]
    host = tester._run([12], sampler.load_gray_batch(paths, 48).to(tester.device), sampler.load_rgb_batch(paths).to(tester.device))
    np.testing.assert_array_equal(host[0], res["utt"].values)
    # Resnet50_Extractor.run writes one %05d.npy per frame and skips when they exist (api/resnet50_extractor.py:61-72)
    out_dir = tmp_path / "utt_pool5"
    tester.resnet50_extractor.run(str(tmp_path / "utt_opface"), str(out_dir), video_name="utt")
    files = sorted(os.listdir(out_dir))
    assert files[0] == "00001.npy" and len(files) == 12 and np.load(str(out_dir / files[0])).shape == (2048,)
    # run() decodes on the host and preprocesses 112 x 112 crops on the GPU: same features, bit for bit, as get_vec on the
    # reference's host-side PIL preprocessing (utils/model_utils.py:29-39)
    want = tester.resnet50_extractor.get_vec(sampler.load_rgb_batch(paths).to(tester.device)).cpu().numpy()
    got = np.stack([np.load(str(out_dir / f)) for f in files])
    np.testing.assert_array_equal(got, want)
    with pytest.raises(RuntimeError, match="aligned faces not found"):
        tester.test(str(tmp_path / "missing.mp4"))


def test_reference_flow_through_the_drop_in_datasets(tester, oracle, tmp_path):
    """The reference's own flow (api/tester.py:53-75) with this build's classes of the same names: Resnet50_Extractor.run writes the
    %05d.npy features, Snippet_Sampler (+ torch DataLoader) yields the windowed snippets, Tester.test_on_dataloader runs them --
    equal to Tester.test's fused path within fp32 regrouping, and to the oracle within the contract.  The uint8 mode of the
    dataset feeds the GPU preprocessing + the de-duplicated phase kernels and gives the fused path's rows bit for bit."""
    from PIL import Image
    from mimamo_net_amd import Snippet_Sampler, Image_Sampler
    n = 40
    clip = synthetic.make_clip_u8(61, n)
    al = tmp_path / "v_opface" / "v_aligned"
    os.makedirs(al)
    for i, f in enumerate(clip):
        Image.fromarray(f, "RGB").save(str(al / ("frame_det_00_%06d.bmp" % (i + 1))))
    feat = tmp_path / "v_pool5"
    tester.resnet50_extractor.run(str(tmp_path / "v_opface"), str(feat), video_name="v")
    ds = Snippet_Sampler("v", str(tmp_path / "v_opface"), str(feat), annot_dir=None, label_name="valence_arousal", test_mode=True,
                         num_phase=12, phase_size=48, length=16, stride=16)
    assert len(ds) == 3 and ds.seq_ranges == [[0, 16], [16, 32], [24, 40]]
    loader = torch.utils.data.DataLoader(ds, batch_size=64, num_workers=0)
    res = tester.test_on_dataloader(loader)["v"].values
    want = _oracle_video(oracle, clip, length=16, stride=16)
    assert res.shape == (n, 2) and np.abs(res - want).max() < OUT_ATOL, np.abs(res - want).max()
    # the raw-boundary mode: unique uint8 frames + window ids per snippet -> GPU preprocessing -> de-duplicated phase kernels
    du = Snippet_Sampler("v", str(tmp_path / "v_opface"), str(feat), label_name="valence_arousal", length=16, stride=16, return_u8=True)
    from mimamo_net_amd.preprocess import FramePreprocessor
    pre = FramePreprocessor(device=tester.device)
    p0s, p1s, rgbs, ranges = [], [], [], []
    with torch.no_grad():
        for k in range(len(du)):
            u8, ids, feats, _, rng, _ = du[k]
            gray, _ = pre(u8.to(tester.device), want_rgb=False)
            p0, p1 = tester.phase_difference_extractor.phase_diff_frames(gray, torch.from_numpy(ids).to(tester.device))
            p0s.append(p0); p1s.append(p1); rgbs.append(torch.from_numpy(feats).to(tester.device)); ranges.append(list(rng))
        out = tester.model([torch.stack(p0s), torch.stack(p1s)], torch.stack(rgbs)).cpu().numpy()     # bs = 3 snippets: GRU over them
    got = sampler.assemble(list(out), ranges, 2)
    assert np.abs(got - want).max() < OUT_ATOL
    assert np.abs(got - res).max() < 2e-5          # windowed (literal kernels) vs de-duplicated (fused kernels) form
    # Image_Sampler hands the extractor the frames it decoded: same features as run() wrote
    im = Image_Sampler("v", str(tmp_path / "v_opface"), test_mode=True, return_u8=True)
    u8 = torch.stack([im[i][0] for i in range(4)]).to(tester.device)
    with torch.no_grad():
        _, rgb3 = pre(u8, want_gray=False, bordered3=True)
        f4 = tester.resnet50_extractor.get_vec(rgb3).cpu().numpy()
    np.testing.assert_array_equal(f4, np.stack([np.load(str(feat / ("%05d.npy" % (i + 1)))) for i in range(4)]))


def test_lanes_on_several_streams_are_bit_identical(tester):
    """Videos spread over HIP streams (shared handles, per-stream workspaces) == single-stream pass."""
    lengths = [64, 100, 64, 30]
    clips = [synthetic.make_clip_u8(70 + i, n) for i, n in enumerate(lengths)]
    frames = torch.from_numpy(np.concatenate(clips)).to(tester.device)
    plan = tester.hot.plan(lengths)
    with torch.no_grad():
        a = tester.hot.forward_u8(frames, plan)
        for lanes in (2, 3):
            b = tester.hot.forward_lanes((frames,), lengths, lanes, from_u8=True)
            torch.cuda.synchronize()
            assert torch.equal(a, b)


def test_bench_sized_step_is_deterministic(tester):
    """BASELINE configs[3] at bench size (32 clips x 64 frames, two lanes): chip-filling launches of every kernel on the
    path are bit-identical run to run and equal to the single-stream pass (load-dependent races show up here first)."""
    dev = tester.device
    g = torch.Generator(device="cpu").manual_seed(5)
    n = 32 * 64
    gray = torch.rand(n, 48, 48, generator=g).to(dev)
    rgb = (torch.rand(n, 224, 224, 4, generator=g) * 200 - 100).to(dev)
    rgb[..., 3] = 0
    lengths = [64] * 32
    with torch.no_grad():
        a = tester.hot.forward_lanes((gray, rgb), lengths, 2, independent_clips=True)
        b = tester.hot.forward_lanes((gray, rgb), lengths, 2, independent_clips=True)
        c = tester.hot.forward(gray, rgb, tester.hot.plan(lengths), independent_clips=True)
        torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b) and torch.equal(a, c)


def test_awkward_video_lengths_alone_and_in_mixed_batches(tester):
    """Videos of 1, 2, 13, 63, 64, 65, 129, 309 frames (short-video rule, tail-snippet overlap, single-frame windows):
    finite, right shapes, and the same values whether a video is processed alone or inside a mixed batch / on lanes."""
    hot, dev = tester.hot, tester.device
    lengths = [1, 2, 13, 63, 64, 65, 129, 309]
    g = torch.Generator(device="cpu").manual_seed(3)
    clips = {n: (torch.rand(n, 48, 48, generator=g).to(dev), (torch.rand(n, 3, 224, 224, generator=g) * 200 - 100).to(dev))
             for n in lengths}
    alone = {}
    with torch.no_grad():
        for n in lengths:
            plan = hot.plan([n])
            res = hot.assemble(hot.forward(*clips[n], plan), plan)[0]
            assert res.shape == (n, 2) and np.isfinite(res).all(), n
            alone[n] = res
        rng = np.random.RandomState(0)
        for it in range(8):
            pick = [lengths[i] for i in rng.randint(0, len(lengths), size=int(rng.randint(2, 5)))]
            gray = torch.cat([clips[n][0] for n in pick])
            rgb = torch.cat([clips[n][1] for n in pick])
            plan = hot.plan(pick)
            out = hot.forward(gray, rgb, plan) if it % 2 else hot.forward_lanes((gray, rgb), pick, 2)
            res = hot.assemble(out, plan)
            for i, n in enumerate(pick):
                assert res[i].shape == (n, 2) and np.abs(res[i] - alone[n]).max() < 1e-5, (pick, i, n)


def test_hot_path_is_hip_graph_capturable(tester):
    """The C ABI only enqueues on the given stream (no hidden synchronisation, allocation or host copy): one pass of the
    whole path can be captured in a HIP graph and replayed on new input values with bit-identical results."""
    dev = tester.device
    g = torch.Generator(device="cpu").manual_seed(9)
    gray = torch.rand(64, 48, 48, generator=g).to(dev)
    rgb = (torch.rand(64, 224, 224, 4, generator=g) * 200 - 100).to(dev)
    plan = tester.hot.plan([64])
    with torch.no_grad():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            tester.hot.forward(gray, rgb, plan, independent_clips=True)     # warm-up outside capture (workspaces)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = tester.hot.forward(gray, rgb, plan, independent_clips=True)
        gray2, rgb2 = gray.flip(0).contiguous(), rgb.flip(0).contiguous()
        want = tester.hot.forward(gray2, rgb2, plan, independent_clips=True).clone()
        gray.copy_(gray2); rgb.copy_(rgb2)          # new values in the captured input buffers
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_reference_style_dataloader_loop_matches_fused_path(tester, oracle):
    """Tester.test_on_dataloader (windowed input, api/tester.py:76-121) == the fused de-duplicated pipeline."""
    n = 100
    clip = synthetic.make_clip_u8(80, n)
    gray, rgb = synthetic.preprocess_host(clip)
    feats = tester.resnet50_extractor.get_vec(torch.from_numpy(rgb).to(tester.device)).cpu().numpy()
    ranges = sampler.snippet_ranges(n)
    batch = (np.stack([gray[sampler.window_ids(s, e, n)] for s, e in ranges]),        # [2,64,13,48,48]
             np.stack([feats[s:e] for s, e in ranges]), None, np.array(ranges), np.array(["v"] * len(ranges)))
    res = tester.test_on_dataloader([batch])
    fused = tester.test_frames([clip], names=["v"])
    assert res["v"].shape == (n, 2)
    # same pyramid per frame; the fused path regroups the blurred ratio per unique frame (csrc/phase_frames.hip): fp32 rounding only
    assert np.abs(res["v"].values - fused["v"].values).max() < 5e-6


def test_equal_shaped_videos_share_one_gru_call_and_match_per_video_calls(tester):
    """plan() lays consecutive videos with the same snippet count out as [snippet][video][frame] and runs ONE head call
    (GRU seq_len = snippets, batch = videos x frames): same bits as one call per video, because GRU batch elements are
    independent (api/mimamo_net.py:119,139) -- and never merges videos of different snippet counts."""
    hot, dev = tester.hot, tester.device
    lengths = [150, 150, 150, 64, 64, 200]
    plan = hot.plan(lengths)
    assert [(g["bs"], g["T"]) for g in plan["groups"]] == [(3, 3 * 64), (1, 2 * 64), (4, 64)]
    g = torch.Generator(device="cpu").manual_seed(11)
    n = sum(lengths)
    gray = torch.rand(n, 48, 48, generator=g).to(dev)
    rgb = (torch.rand(n, 224, 224, 4, generator=g) * 200 - 100).to(dev)
    rgb[..., 3] = 0
    with torch.no_grad():
        res = hot.assemble(hot.forward(gray, rgb, plan), plan)
        off = 0
        for i, L in enumerate(lengths):
            p1 = hot.plan([L])
            assert len(p1["groups"]) == 1
            alone = hot.assemble(hot.forward(gray[off:off + L].contiguous(), rgb[off:off + L].contiguous(), p1), p1)[0]
            np.testing.assert_array_equal(res[i], alone)
            off += L


def test_long_videos_stream_through_bounded_chunks(pkg, oracle):
    """A video longer than batch_size snippets is split into the reference's DataLoader batches (api/tester.py:69-72), each
    its own GRU call; preprocessing + ResNet50 run max_frames_per_call frames at a time.  Same values as the oracle run
    with the same batch size, and as an un-chunked pass."""
    from mimamo_net_amd.pipeline import HotPath
    head_sd, rs_sd = weights.make_two_stream_state_dict(seed=0), weights.make_resnet50_state_dict(seed=0)
    small = HotPath(head_sd, rs_sd, "cuda:0", length=8, stride=8, batch_size=2, max_frames_per_call=7)
    big = HotPath(head_sd, rs_sd, "cuda:0", length=8, stride=8, batch_size=2)
    n = 37                                           # 5 snippets (tail [29,37)) -> batches of 2, 2, 1 snippets
    clip = synthetic.make_clip_u8(90, n)
    plan = small.plan([n])
    assert [(g["bs"], g["T"]) for g in plan["groups"]] == [(2, 8), (2, 8), (1, 8)]
    frames = torch.from_numpy(clip).to("cuda:0")
    with torch.no_grad():
        a = small.assemble(small.forward_u8(frames, plan), plan)[0]
        b = big.assemble(big.forward_u8(frames, big.plan([n])), plan)[0]
    np.testing.assert_array_equal(a, b)
    # oracle with the reference's semantics at the same snippet length / batch size
    gray, rgb = synthetic.preprocess_host(clip)
    ranges = oracle.snippet_ranges(n, 8, 8)
    feats = oracle.resnet50_pool5(rs_sd, rgb)
    ph = np.stack([gray[oracle.window_ids(s, e, n)] for s, e in ranges])
    rg = np.stack([feats[s:e] for s, e in ranges])
    preds = []
    for c0 in range(0, len(ranges), 2):
        p0, p1 = oracle.phase_diff_output(ph[c0:c0 + 2])
        preds.extend(list(oracle.two_stream_forward(head_sd, p0, p1, rg[c0:c0 + 2])))
    want = oracle.assemble(preds, ranges)
    assert np.abs(a - want).max() < OUT_ATOL


def test_plan_and_window_ids_are_validated(tester):
    hot, dev = tester.hot, tester.device
    plan = hot.plan([64])
    with pytest.raises(ValueError, match="plan was built for 64 frames"):
        hot.forward(torch.zeros(32, 48, 48, device=dev), torch.zeros(32, 224, 224, 4, device=dev), plan)
    ids = torch.from_numpy(sampler.window_ids(0, 64, 64)).to(dev)
    with pytest.raises(ValueError, match="window_ids must index"):
        tester.phase_difference_extractor.phase_diff_frames(torch.rand(32, 48, 48, device=dev), ids)


def test_host_resident_frames_stream_in_chunks_with_identical_rows(tester):
    """HotPath.forward_u8 on a pinned HOST tensor: uploaded chunk by chunk on the copy stream under the previous chunk's
    compute (stream.FrameStream) -- bit-identical to the device-resident call, for a chunk size that does not divide the
    video and for one larger than it."""
    from mimamo_net_amd.stream import pin
    clips = [synthetic.make_clip_u8(70, 150), synthetic.make_clip_u8(71, 64)]
    host = pin(np.concatenate(clips))
    assert host.is_pinned() and not host.is_cuda
    plan = tester.hot.plan([150, 64])
    keep = tester.hot.upload_chunk_frames
    try:
        with torch.no_grad():
            want = tester.hot.forward_u8(host.to(tester.device), plan)
            for chunk in (64, 100, 4096):
                tester.hot.upload_chunk_frames = chunk
                got = tester.hot.forward_u8(host, plan)
                assert torch.equal(got, want), chunk
            # pageable source: still correct (the copy is then synchronous with the host)
            got = tester.hot.forward_u8(torch.from_numpy(np.concatenate(clips)), plan)
            assert torch.equal(got, want)
    finally:
        tester.hot.upload_chunk_frames = keep


def test_tester_with_a_non_published_pyramid_configuration(golden, oracle):
    """Tester forwards its sampler / pyramid keywords (api/tester.py:15-33): a 4-band pyramid over 7-frame windows still hands
    the default Two_Stream_RNN() its 24 channels per level (api/tester.py:44-45), so it runs -- in the reference and here, on the
    general pyramid + generic extract kernels.  (a) phase_diff_output + model against the real reference's outputs
    (tests/golden/nondefault.npz); (b) a whole video through Tester.test_frames against the oracle with the same keywords;
    (c) a configuration whose channel count does not fit the model fails at the model's input check."""
    from mimamo_net_amd.tester import Tester
    from mimamo_net_amd.phase_difference_extractor import phase_diff_output
    g = golden("nondefault")
    head_sd = weights.make_two_stream_state_dict(seed=int(g["weight_seed_b"]))
    resnet_sd = weights.make_resnet50_state_dict(seed=0)
    t = Tester(model_path=None, batch_size=64, head_state_dict=head_sd, resnet_state_dict=resnet_sd, device="cuda:0",
               num_phase=6, nbands=4)
    assert not t.hot.fused
    clip = synthetic.textured_gray(10, 48, seed=61)
    w = np.stack([clip[i:i + 7] for i in range(4)])[None]
    q0, q1 = phase_diff_output(torch.from_numpy(w).to(t.device), t.phase_difference_extractor)
    for got, want in ((q0, g["b_phase_0"]), (q1, g["b_phase_1"])):
        d = np.abs(got.cpu().numpy()[0] - want)
        print("4-band / 7-frame windows vs real reference: max %.2e" % d.max())
        assert d.max() < 1e-3 and np.quantile(d, 0.9999) < 3e-4 and d.max() < 1e-4       # contract, then regression bound
    rgb_b = torch.from_numpy(weights.det_uniform("nd.rgb_b", (1, 4, 2048), 0.0, 2.0, 72)).to(t.device)
    y = t.model([q0, q1], rgb_b).cpu().numpy()
    assert np.abs(y - g["b_out"]).max() < OUT_ATOL
    # (b) a 70-frame video: snippets [0,64) and [6,70), windows of 7 clamped frames
    video = synthetic.make_clip_u8(80, 70)
    res = t.test_frames([video], names=["v"])["v"].values
    gray, rgb = synthetic.preprocess_host(video)
    n = len(video)
    ranges = oracle.snippet_ranges(n, 64, 64)
    feats = oracle.resnet50_pool5(resnet_sd, rgb)
    ph = np.stack([gray[oracle.window_ids(s, e, n, 6)] for s, e in ranges])
    p0, p1 = oracle.phase_diff_output(ph, height=4, nbands=4)
    want = oracle.assemble(list(oracle.two_stream_forward(head_sd, p0, p1, np.stack([feats[s:e] for s, e in ranges]))), ranges)
    err = np.abs(res - want).max()
    print("Tester(num_phase=6, nbands=4) 70-frame video vs oracle: %.2e" % err)
    assert res.shape == (70, 2) and err < OUT_ATOL
    # (c) 2 bands x 6 differences = 12 channels: the default model refuses them
    t2 = Tester(model_path=None, batch_size=64, head_state_dict=head_sd, resnet_state_dict=resnet_sd, device="cuda:0", num_phase=6)
    with pytest.raises(AssertionError):
        t2.test_frames([video[:16]])
