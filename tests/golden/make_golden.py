"""Freeze outputs of the REAL reference (imported from /root/reference via ref_shim) as small
data fixtures.  Run once in the build container:  python tests/golden/make_golden.py

Fixtures are data only (inputs/expected outputs); no reference source is stored.
Inputs are regenerated in the tests from the repo's own closed-form generators
(mimamo-net_amd/weights.py det_uniform, synthetic.textured_gray), so only the generator
arguments and the reference's outputs are saved.

  G1 masks.npz      every mask pointOp produced inside SCFpyr_PyTorch.build for a 96x96 input,
                    + the crop offsets actually used (located inside the parent grid)
  G2 pyramid.npz    build_pyramid of 4 textured 48x48 frames: fp32 and fp64 reference, + the
                    independent SCFpyr_NumPy result for frame 0
  G3 extract.npz    Tester.phase_diff_output on 3 windows (moving / edge-clamped / fast-moving)
  KAT kats.npz      torch_unwrap, torch_diff, gaussian_kernel, symmetric_extension_batch, blur
  G4 head.npz       Two_Stream_RNN(eval) with generated weights, bs in {1,3}, T=4
  G8 scfpyr_full.npz  SCFpyr_PyTorch.build (the FULL list: hi-pass residual, every band, low-pass residual) on
                    non-symmetric images: 96x96 height 4 / 2 bands (stored fp32, computed at precision=64) and
                    32x32 height 3 with 4 and 3 bands (float64; complex factors (-i)^3 and (-i)^2)
  G5 resnet50_hf.npz  NOT from the reference (its ResNet50 lives in an un-vendored third-party file): pool5 features of an
                    INDEPENDENT third-party implementation of the same Caffe-style graph -- Hugging Face transformers
                    ResNetModel(downsample_in_bottleneck=True), pooler swapped for MaxPool2d(3, 2, 0, ceil_mode=True) --
                    loaded with the build's deterministic weights; pins the executor's wiring (stride placement,
                    projection shortcuts, BN folding, pooling), not the unavailable checkpoint
  G9 phase_generic.npz  Phase_Difference_Extractor with OTHER constructor arguments than api/tester.py's: (height 3, 4 bands,
                    level 1, 5 textured 32x32 frames, symmetry) and (height 3, 2 bands, level [1], 3 frames, symmetry=False),
                    and the class defaults (height 5, 4 bands, level 1) on 64x64 frames: build_pyramid -> extract outputs
  G10 train_phase.npz  training-side Steerable_Pyramid_Phase (Aff-wild-exps/utils.py:298-418): build_pyramid in fp32, then
                    extract_phase(default / return_phase / return_both) on the float64 cast of those coefficients (the
                    class's blur only casts its kernel to float32 on CUDA, so on this CPU-only host it runs in float64)
  G7 sampler.npz    Snippet_Sampler.seq_ranges for N in {10,64,100,128,309} and the 13-frame window
                    ids decoded from constant-valued BMPs, + one textured BMP pass pinning
                    convert('L') + Lanczos 112->48 + /255
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
import mimamo_net_amd  # noqa: E402,F401
from mimamo_net_amd import weights, synthetic  # noqa: E402


def g1_masks(ref):
    mod = ref.scf_torch_module
    rec = []
    orig = mod.pointOp

    def spy(im, y, x):
        out = orig(im, y, x)
        rec.append((np.array(im, copy=True), np.array(out, copy=True)))
        return out

    mod.pointOp = spy
    try:
        pyr = ref.SCFpyr_PyTorch(height=4, nbands=2, scale_factor=2, device=torch.device("cpu"))
        torch.set_default_dtype(torch.float32)
        x = torch.from_numpy(synthetic.textured_gray(1, 96, seed=3))[:, None]
        pyr.build(x)
    finally:
        mod.pointOp = orig
    # call order: lo0, hi0, [himask, angle0, angle1, lomask] x 2 levels
    assert len(rec) == 2 + 4 * 2, len(rec)
    out = {"lo0": rec[0][1], "hi0": rec[1][1]}
    grid = rec[0][0]
    for l in range(2):
        him, a0, a1, lom = rec[2 + 4 * l: 6 + 4 * l]
        out["himask_%d" % l] = him[1]
        out["anglemask_%d_0" % l] = a0[1]
        out["anglemask_%d_1" % l] = a1[1]
        out["lomask_%d" % l] = lom[1]
        # locate the cropped log_rad grid inside its parent -> crop offsets the reference used
        sub, n = lom[0], lom[0].shape[0]
        found = [(i, j) for i in range(grid.shape[0] - n + 1) for j in range(grid.shape[1] - n + 1)
                 if np.array_equal(grid[i:i + n, j:j + n], sub)]
        assert len(found) == 1, found
        out["crop_%d" % l] = np.array([found[0][0], found[0][0] + n, found[0][1], found[0][1] + n])
        grid = sub
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)
    print("G1", {k: v.shape for k, v in out.items()}, out["crop_0"], out["crop_1"])


def g2_pyramid(ref):
    frames = synthetic.textured_gray(4, 48, seed=11)
    x = torch.from_numpy(frames)[None]  # [1,4,48,48]
    pde = ref.Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
    torch.set_default_dtype(torch.float32)
    c32 = [c.numpy() for c in pde.build_pyramid(x)]
    pde64 = ref.Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
    pde64.pyramid = ref.SCFpyr_PyTorch(4, 2, 2, device=torch.device("cpu"), precision=64)
    c64 = [c.numpy() for c in pde64.build_pyramid(x.double())]
    torch.set_default_dtype(torch.float32)
    # independent numpy implementation, single mirrored image
    sym = ref.phase_utils.symmetric_extension_batch(x.view(4, 1, 48, 48))[0, 0].double().numpy()
    cn = ref.SCFpyr_NumPy(height=4, nbands=2, scale_factor=2, precision=64).build(sym)
    np.savez_compressed(os.path.join(HERE, "pyramid.npz"),
                        seed=11, l1_f32=c32[0], l2_f32=c32[1], l1_f64=c64[0], l2_f64=c64[1],
                        np_l1=np.stack([cn[1][0][:48, :48], cn[1][1][:48, :48]]),
                        np_l2=np.stack([cn[2][0][:24, :24], cn[2][1][:24, :24]]))
    print("G2", c32[0].shape, c32[1].shape, np.abs(c32[0] - c64[0]).max(), np.abs(c32[1] - c64[1]).max())


def make_windows():
    """3 windows [13,48,48]: smooth motion; clamped at the video start (frames 0..6 replicated);
    fast motion (large inter-frame phase steps -> negative jumps that fmod-unwrap leaves)."""
    slow = synthetic.textured_gray(13, 48, seed=21)
    ids = np.clip(np.arange(13) - 6, 0, None)
    clamped = synthetic.textured_gray(7, 48, seed=22)[ids]
    fast = synthetic.textured_gray(13 * 5, 48, seed=23)[::5]
    return np.stack([slow, clamped, fast])[None]  # [1,3,13,48,48]


def g3_extract(ref):
    w = make_windows()
    pde = ref.Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
    torch.set_default_dtype(torch.float32)
    p0, p1 = ref.Tester.phase_diff_output(None, torch.from_numpy(w), pde)
    pde64 = ref.Phase_Difference_Extractor(4, 2, 2, [1, 2], False)
    pde64.pyramid = ref.SCFpyr_PyTorch(4, 2, 2, device=torch.device("cpu"), precision=64)
    q0, q1 = ref.Tester.phase_diff_output(None, torch.from_numpy(w).double(), pde64)
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, "extract.npz"), phase_0=p0.numpy()[0], phase_1=p1.numpy()[0],
                        phase_0_f64=q0.numpy()[0].astype(np.float32), phase_1_f64=q1.numpy()[0].astype(np.float32))
    d0 = np.abs(p0.numpy() - q0.numpy())
    print("G3", p0.shape, p1.shape, "fp32-vs-fp64 max", d0.max(), "p99.99", np.quantile(d0, 0.9999))


def kats(ref):
    pu = ref.phase_utils
    a = torch.tensor([0.0, 3.0, -3.0, 3.0, -3.0])
    r = weights.det_uniform("kat.unwrap", (6, 13, 5), -3.1415926, 3.1415926, 5)
    r[0, :, 0] = np.array([0, 3, -3, 3, -3, 0.5, 3.1, -3.1, 2, -2, 3, 0, -3], dtype=np.float32)
    # exact +-pi jumps exercise the `ddmod == -pi & dd > 0` branch
    r[1, :, 1] = np.float32(np.pi) * np.array([0, 1, 0, -1, 0, 1, 2, 1, 0, -1, -2, -1, 0], dtype=np.float32)
    out = {
        "unwrap5_in": a.numpy(), "unwrap5_out": pu.torch_unwrap(a, dim=-1).numpy(),
        "unwrap_in": r, "unwrap_out": pu.torch_unwrap(torch.from_numpy(r), dim=-2).numpy(),
        "diff_out": pu.torch_diff(torch.from_numpy(r), dim=1).numpy(),
        "gauss": pu.gaussian_kernel(std=2, tap=11),
    }
    s = weights.det_uniform("kat.sym", (2, 1, 4, 6), 0, 1, 5)
    out["sym_in"] = s
    out["sym_out"] = pu.symmetric_extension_batch(torch.from_numpy(s)).numpy()
    mag = weights.det_uniform("kat.mag", (2, 3, 12, 12), 0.01, 1.0, 5)
    ph = weights.det_uniform("kat.ph", (2, 3, 12, 12), -6.0, 6.0, 5)
    out["blur_mag"], out["blur_ph"] = mag, ph
    out["blur_out"] = pu.amplitude_based_gaussian_blur(torch.from_numpy(mag), torch.from_numpy(ph),
                                                       torch.from_numpy(pu.gaussian_kernel(2, 11))).numpy()
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **out)
    print("KAT unwrap5", out["unwrap5_out"], "gauss sum", out["gauss"].sum())


def head_inputs(bs, t, seed):
    p0 = weights.det_uniform("head.p0", (bs, t, 24, 48, 48), -1.5, 1.5, seed)
    p1 = weights.det_uniform("head.p1", (bs, t, 24, 24, 24), -1.5, 1.5, seed)
    rgb = weights.det_uniform("head.rgb", (bs, t, 2048), 0.0, 2.0, seed)
    return p0, p1, rgb


def g4_head(ref):
    sd = weights.make_two_stream_state_dict(seed=3)
    model = ref.Two_Stream_RNN()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    out = {"weight_seed": 3}
    with torch.no_grad():
        for bs in (1, 3):
            p0, p1, rgb = head_inputs(bs, 4, 40 + bs)
            y = model([torch.from_numpy(p0), torch.from_numpy(p1)], torch.from_numpy(rgb)).numpy()
            out["out_bs%d" % bs] = y
            out["in_seed_bs%d" % bs] = 40 + bs
            print("G4 bs", bs, y.reshape(-1, 2)[:3])
    np.savez_compressed(os.path.join(HERE, "head.npz"), **out)


def g7_sampler(ref):
    from PIL import Image
    out = {}
    for n in (10, 64, 100, 128, 309):
        with tempfile.TemporaryDirectory() as d:
            feat = os.path.join(d, "feat")
            root = os.path.join(d, "v_opface")
            os.makedirs(feat)
            os.makedirs(os.path.join(root, "v_aligned"))
            for i in range(1, n + 1):
                np.save(os.path.join(feat, "%05d.npy" % i), np.full((4,), i, dtype=np.float32))
                val = (i - 1) % 251  # constant image encodes the 0-based frame id
                Image.fromarray(np.full((16, 16, 3), val, dtype=np.uint8), "RGB").save(
                    os.path.join(root, "v_aligned", "frame_det_00_%06d.bmp" % i))
            ds = ref.Snippet_Sampler("v", root, feat, annot_dir=None, label_name="valence_arousal",
                                     test_mode=True, num_phase=12, phase_size=8, length=64, stride=64)
            out["ranges_%d" % n] = np.array(ds.seq_ranges)
            ids = []
            for k in range(len(ds)):
                ph, feats, _, rng, _ = ds[k]
                ids.append(np.rint(ph[:, :, 0, 0].numpy() * 255).astype(np.int64))
                assert np.array_equal(feats[:, 0].astype(np.int64) - 1, np.arange(rng[0], rng[1]))
            out["ids_%d" % n] = np.stack(ids)  # [n_snip, len, 13] (frame id mod 251)
    # preprocessing pin: textured 112x112 frames through the real sampler (convert L + Lanczos + /255)
    clip = synthetic.make_clip_u8(5, 3)
    with tempfile.TemporaryDirectory() as d:
        feat = os.path.join(d, "feat")
        root = os.path.join(d, "v_opface")
        os.makedirs(feat)
        os.makedirs(os.path.join(root, "v_aligned"))
        for i in range(1, 4):
            np.save(os.path.join(feat, "%05d.npy" % i), np.zeros((4,), dtype=np.float32))
            Image.fromarray(clip[i - 1], "RGB").save(os.path.join(root, "v_aligned", "frame_det_00_%06d.bmp" % i))
        ds = ref.Snippet_Sampler("v", root, feat, annot_dir=None, label_name="valence_arousal",
                                 test_mode=True, num_phase=12, phase_size=48, length=64, stride=64)
        ph = ds[0][0].numpy()  # [3,13,48,48]
        out["gray48_clip5"] = np.stack([ph[0, 6], ph[1, 6], ph[2, 6]])
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("G7", {k: v.shape for k, v in out.items()})


def resnet_images(n, seed):
    """[n,3,224,224] in the extractor's input convention (255*x - mean, utils/model_utils.py:36-39)."""
    x = weights.det_uniform("resnet.img", (n, 3, 224, 224), 0.0, 1.0, seed)
    return (x * np.float32(255.0) - np.asarray(weights.RESNET50_MEAN, dtype=np.float32)[None, :, None, None]).astype(np.float32)


def g5_resnet50_hf():
    """Graph wiring against code this build did not write: Hugging Face transformers' ResNetModel with the build's deterministic
    weights.  Variant (stride_on_first_1x1, ceil_mode): (1, 1) is the published Caffe-style graph (keys pool5_f32 / pool5_f64);
    (0, 1), (1, 0), (0, 0) pin the other settings of the two knobs SURVEY 8(c) keeps open (keys pool5_f64_v01 ...)."""
    import transformers
    from transformers import ResNetConfig, ResNetModel
    sd = weights.make_resnet50_state_dict(seed=0)

    def build(stride_on_first_1x1, ceil_mode):
        model = ResNetModel(ResNetConfig(downsample_in_bottleneck=bool(stride_on_first_1x1))).eval()
        model.embedder.pooler = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=0, ceil_mode=bool(ceil_mode))   # pool1_3x3_s2
        hf = {}

        def put(dst, src):
            hf[dst + ".convolution.weight"] = torch.from_numpy(sd[src + ".weight"])
            for k in ("weight", "bias", "running_mean", "running_var"):
                hf[dst + ".normalization." + k] = torch.from_numpy(sd[src + "_bn." + k])

        put("embedder.embedder", "conv1_7x7_s2")
        for si, (stage, blocks, _, _, _) in enumerate(weights.RESNET50_STAGES):
            for b in range(1, blocks + 1):
                pre, dst = "conv%d_%d_" % (stage, b), "encoder.stages.%d.layers.%d" % (si, b - 1)
                if b == 1:
                    put(dst + ".shortcut", pre + "1x1_proj")
                put(dst + ".layer.0", pre + "1x1_reduce")
                put(dst + ".layer.1", pre + "3x3")
                put(dst + ".layer.2", pre + "1x1_increase")
        missing, unexpected = model.load_state_dict(hf, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
        return model

    out = {"transformers_version": transformers.__version__, "weight_seed": 0}
    for (s1, cm), tag in (((1, 1), ""), ((0, 1), "_v01"), ((1, 0), "_v10"), ((0, 0), "_v00")):
        model = build(s1, cm)
        for prec, dt in (("f32", torch.float32), ("f64", torch.float64)):
            if tag and prec == "f32":
                continue
            m = model.to(dt)
            with torch.no_grad():
                y = m(torch.from_numpy(resnet_images(2, 7)).to(dt)).pooler_output     # [2,2048,1,1] = AdaptiveAvgPool of stage 4
            out["pool5_" + prec + tag] = y.reshape(2, 2048).numpy()
    print("G5", out["pool5_f32"].shape, np.abs(out["pool5_f32"]).max(), np.abs(out["pool5_f32"] - out["pool5_f64"]).max(),
          [float(np.abs(out["pool5_f64" + t] - out["pool5_f64"]).max()) for t in ("_v01", "_v10", "_v00")])
    np.savez_compressed(os.path.join(HERE, "resnet50_hf.npz"), **out)


def g9_phase_generic(ref):
    out = {}
    x = torch.from_numpy(synthetic.textured_gray(5, 32, seed=31))[None]            # [1,5,32,32]
    pde = ref.Phase_Difference_Extractor(3, 4, 2, 1, False)
    torch.set_default_dtype(torch.float32)
    c = pde.build_pyramid(x)                                                        # [1,4,5,32,32,2]
    out["a_coeff"] = c.numpy()
    out["a_diff"] = pde.extract(c).numpy()                                          # [1,4,4,32,32]
    x2 = torch.from_numpy(synthetic.textured_gray(3, 32, seed=32))[None]
    pde2 = ref.Phase_Difference_Extractor(3, 2, 2, [1], False)
    torch.set_default_dtype(torch.float32)
    c2 = pde2.build_pyramid(x2, symmetry=False)                                     # [[1,2,3,32,32,2]]
    out["b_coeff"] = c2[0].numpy()
    out["b_diff"] = pde2.extract(c2[0]).numpy()
    # (c) the class DEFAULTS (height 5, 4 bands, level 1) need >= 64x64 frames: mirrored side 128
    x3 = torch.from_numpy(synthetic.textured_gray(3, 64, seed=33))[None]
    pde3 = ref.Phase_Difference_Extractor()
    torch.set_default_dtype(torch.float32)
    c3 = pde3.build_pyramid(x3)                                                     # [1,4,3,64,64,2]
    out["c_coeff_band0"] = c3[:, :1].numpy()
    out["c_diff"] = pde3.extract(c3).numpy()                                        # [1,4,2,64,64]
    torch.set_default_dtype(torch.float32)
    print("G9", {k: v.shape for k, v in out.items()})
    np.savez_compressed(os.path.join(HERE, "phase_generic.npz"), **out)


def g10_train_phase(ref):
    import types
    path = "/root/reference/Aff-wild-exps/utils.py"
    src = open(path).read().replace("async=True", "non_blocking=True")     # same py<3.7 keyword as shim 4
    mod = types.ModuleType("affwild_utils")
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)
    sp = mod.Steerable_Pyramid_Phase(height=4, nbands=2, scale_factor=2, device=torch.device("cpu"), extract_level=[1, 2],
                                     visualize=False)
    torch.set_default_dtype(torch.float32)
    x = torch.from_numpy(synthetic.textured_gray(6, 48, seed=41))[None]           # [1,6,48,48]
    c1, c2 = sp.build_pyramid(x)
    out = {"c1": c1.numpy(), "c2": c2.numpy()}
    for tag, c in (("l1", c1), ("l2", c2)):
        cd = c.double()
        out[tag + "_diff"] = sp.extract_phase(cd).numpy().astype(np.float32)
        out[tag + "_phase"] = sp.extract_phase(cd, return_phase=True).numpy().astype(np.float32)
        sp_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self                          # return_both ends with result.cuda() (:416)
        try:
            out[tag + "_both"] = sp.extract_phase(cd, return_both=True).numpy().astype(np.float32)
        finally:
            torch.Tensor.cuda = sp_cuda
    torch.set_default_dtype(torch.float32)
    print("G10", {k: v.shape for k, v in out.items()})
    np.savez_compressed(os.path.join(HERE, "train_phase.npz"), **out)


SCF_FULL_CASES = [
    # tag, size, height, nbands, n_images, seed, stored dtype
    ("a", 96, 4, 2, 1, 8, np.float32),
    ("b", 32, 3, 4, 2, 9, np.float64),
    ("c", 32, 3, 3, 1, 10, np.float64),
    # round 5: odd grids -- 50 -> 25 (odd level), 75 (odd image) -> 38 -> 19, 84 -> 42 -> 21 (rejected until round 5)
    ("d", 50, 3, 2, 1, 11, np.float64),
    ("e", 75, 4, 2, 1, 12, np.float64),
    ("f", 84, 4, 2, 1, 13, np.float32),
]


def g8_scfpyr_full(ref):
    out = {}
    for tag, size, height, nbands, n, seed, dt in SCF_FULL_CASES:
        x = weights.det_uniform("scf." + tag, (n, 1, size, size), 0.0, 1.0, seed)
        pyr = ref.SCFpyr_PyTorch(height=height, nbands=nbands, scale_factor=2, device=torch.device("cpu"), precision=64)
        coeff = pyr.build(torch.from_numpy(x).double())
        torch.set_default_dtype(torch.float32)
        assert len(coeff) == height and isinstance(coeff[1], list) and len(coeff[1]) == nbands
        out["%s_hi" % tag] = coeff[0].numpy().astype(dt)
        out["%s_lo" % tag] = coeff[-1].numpy().astype(dt)
        for l in range(1, height - 1):
            out["%s_l%d" % (tag, l)] = np.stack([b.numpy() for b in coeff[l]]).astype(dt)  # [nbands, N, s, s, 2]
        print("G8", tag, {k: v.shape for k, v in out.items() if k.startswith(tag)})
    np.savez_compressed(os.path.join(HERE, "scfpyr_full.npz"), **out)


def g12_sampler_keywords(ref):
    """Snippet_Sampler with other length / stride / num_phase keywords (api/sampler/snippet_sampler.py:107-152): overlapping
    snippets (stride < length), gaps (stride > length), a short video, one- and two-frame videos, an even number of frames in
    the window -- ranges and clamped window ids of the real class."""
    from PIL import Image
    out = {}
    cases = [(100, 32, 16, 12), (309, 64, 32, 12), (50, 64, 64, 12), (65, 64, 64, 12), (1, 64, 64, 12), (2, 64, 64, 12), (13, 8, 5, 12),
             (40, 16, 24, 6), (30, 10, 10, 7)]
    for n, length, stride, num_phase in cases:
        with tempfile.TemporaryDirectory() as d:
            feat = os.path.join(d, "feat")
            root = os.path.join(d, "v_opface")
            os.makedirs(feat)
            os.makedirs(os.path.join(root, "v_aligned"))
            for i in range(1, n + 1):
                np.save(os.path.join(feat, "%05d.npy" % i), np.full((4,), i, dtype=np.float32))
                Image.fromarray(np.full((8, 8, 3), (i - 1) % 251, dtype=np.uint8), "RGB").save(
                    os.path.join(root, "v_aligned", "frame_det_00_%06d.bmp" % i))
            ds = ref.Snippet_Sampler("v", root, feat, annot_dir=None, label_name="valence_arousal", test_mode=True,
                                     num_phase=num_phase, phase_size=8, length=length, stride=stride)
            tag = "%d_%d_%d_%d" % (n, length, stride, num_phase)
            out["ranges_" + tag] = np.array(ds.seq_ranges)
            ids = []
            for k in range(len(ds)):
                ph = ds[k][0]
                ids.append(np.rint(ph[:, :, 0, 0].numpy() * 255).astype(np.int64))
            out["ids_" + tag] = np.stack(ids)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "sampler_keywords.npz"), **out)
    print("G12", {k: v.shape for k, v in out.items() if k.startswith("ranges")})


def nondefault_windows(T=4, P=7):
    """[1,T,P,48,48]: T consecutive 7-frame windows of one textured clip (num_phase = 6)."""
    clip = synthetic.textured_gray(T + P - 1, 48, seed=61)
    return np.stack([clip[t:t + P] for t in range(T)])[None]


def g11_nondefault(ref):
    """A configuration other than api/tester.py:28-32's, through the real reference:
    (a) Two_Stream_RNN(num_phase=6): PhaseNet with 12 input channels per level (api/mimamo_net.py:97-112);
    (b) the Tester's chain with a 4-band pyramid and 7-frame windows -- nbands * num_phase = 24 channels at 48x48 and 24x24,
        which is what its default Two_Stream_RNN() takes (api/tester.py:28-45,122-139): phase_diff_output + model."""
    out = {"weight_seed_a": 5, "weight_seed_b": 3}
    torch.set_default_dtype(torch.float32)
    sd = weights.make_two_stream_state_dict(seed=5, num_phase=6)
    model = ref.Two_Stream_RNN(num_phase=6)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    p0 = weights.det_uniform("nd.p0", (2, 3, 12, 48, 48), -1.5, 1.5, 71)
    p1 = weights.det_uniform("nd.p1", (2, 3, 12, 24, 24), -1.5, 1.5, 71)
    rgb = weights.det_uniform("nd.rgb", (2, 3, 2048), 0.0, 2.0, 71)
    with torch.no_grad():
        out["a_out"] = model([torch.from_numpy(p0), torch.from_numpy(p1)], torch.from_numpy(rgb)).numpy()
    # (c) an odd num_phase (2 * 5 = 10 channels: not a multiple of the build's 4-channel groups) and one beyond 32
    for tag, nph, seed in (("c5", 5, 6), ("c40", 40, 7)):
        sdc = weights.make_two_stream_state_dict(seed=seed, num_phase=nph)
        mc = ref.Two_Stream_RNN(num_phase=nph)
        mc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sdc.items()}, strict=True)
        mc.eval()
        c0 = weights.det_uniform("nd.%s.p0" % tag, (2, 2, 2 * nph, 48, 48), -1.5, 1.5, 73)
        c1 = weights.det_uniform("nd.%s.p1" % tag, (2, 2, 2 * nph, 24, 24), -1.5, 1.5, 73)
        crgb = weights.det_uniform("nd.%s.rgb" % tag, (2, 2, 2048), 0.0, 2.0, 73)
        with torch.no_grad():
            out[tag + "_out"] = mc([torch.from_numpy(c0), torch.from_numpy(c1)], torch.from_numpy(crgb)).numpy()
        out["weight_seed_" + tag] = seed
    w = nondefault_windows()
    pde = ref.Phase_Difference_Extractor(4, 4, 2, [1, 2], False)
    q0, q1 = ref.Tester.phase_diff_output(None, torch.from_numpy(w), pde)
    torch.set_default_dtype(torch.float32)
    sd = weights.make_two_stream_state_dict(seed=3)
    model = ref.Two_Stream_RNN()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    rgb_b = weights.det_uniform("nd.rgb_b", (1, w.shape[1], 2048), 0.0, 2.0, 72)
    with torch.no_grad():
        out["b_out"] = model([q0.float(), q1.float()], torch.from_numpy(rgb_b)).numpy()
    out["b_phase_0"] = q0.numpy()[0].astype(np.float32)
    out["b_phase_1"] = q1.numpy()[0].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "nondefault.npz"), **out)
    print("G11", out["a_out"].reshape(-1, 2)[:2], q0.shape, q1.shape, out["b_out"].reshape(-1, 2)[:2])


def g13_resnet50_extractor_plumbing(ref):
    """The reference's OWN code on the ResNet50 row, executed for the first time: `Resnet50_Extractor(benchmark_dir, model_name,
    'pool5_7x7_s1')` (api/resnet50_extractor.py:14-41: `load_model`'s importlib exec of `<benchmark_dir>/ferplus/<model_name>.py` and its
    `weights_path=` factory call, api/utils/model_utils.py:44-79; `.eval()`; `meta` -> `compose_transforms`) and `.get_vec`
    (:74-83: forward hook on `_modules['pool5_7x7_s1']`, copy into a [bs,2048,1,1] CPU tensor, `relu(squeeze())`) on a STAND-IN
    definition file + deterministic weights (tests/golden/standin_model.py -- this build's restatement of the graph, the third-party
    file is not available offline).  Pins the plumbing (layer name, hook, squeeze, meta, key layout of the .pth), not the arithmetic."""
    import tempfile
    import standin_model
    import resnet50_extractor as rex
    tmp = tempfile.mkdtemp(prefix="mm_g13_")
    bdir, _ = standin_model.write_benchmark_dir(os.path.join(tmp, "pytorch-benchmarks"), weights, seed=5)
    ext = rex.Resnet50_Extractor(benchmark_dir=bdir, model_name="resnet50_ferplus_dag", feature_layer="pool5_7x7_s1")
    assert not ext.model.training
    x = resnet_images(2, 13)
    with torch.no_grad():
        v2 = ext.get_vec(torch.from_numpy(x))
        v1 = ext.get_vec(torch.from_numpy(x[:1]))                       # quirk Q8: squeeze() collapses bs = 1 to [2048]
    out = {"weight_seed": 5, "image_seed": 13, "vec_bs2": v2.numpy(), "vec_bs1": v1.numpy(),
           "meta_mean": np.asarray(ext.model.meta["mean"], dtype=np.float64), "meta_std": np.asarray(ext.model.meta["std"]),
           "meta_imageSize": np.asarray(ext.model.meta["imageSize"]),
           "frame_index": np.asarray([ext.get_frame_index("/a/b_aligned/frame_det_00_000123.bmp")]),
           "hooked_layer_type": np.asarray(type(ext.model._modules.get(ext.feature_layer)).__name__),
           "n_modules": np.asarray(len(ext.model._modules))}
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print("G13", out["vec_bs2"].shape, out["vec_bs1"].shape, float(np.abs(out["vec_bs2"]).max()), out["hooked_layer_type"], out["n_modules"])
    np.savez_compressed(os.path.join(HERE, "resnet50_plumbing.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] in ([], ["g5"]):
        g5_resnet50_hf()   # before ref_shim.load(): its torchvision stub confuses transformers' optional-dependency probe
        if sys.argv[1:]:
            sys.exit(0)
    ref = ref_shim.load()
    if sys.argv[1:] == ["g13"]:
        g13_resnet50_extractor_plumbing(ref)
        sys.exit(0)
    if sys.argv[1:] == ["g12"]:
        g12_sampler_keywords(ref)
        sys.exit(0)
    if sys.argv[1:] == ["g11"]:
        g11_nondefault(ref)
        sys.exit(0)
    if sys.argv[1:] == ["g10"]:
        g10_train_phase(ref)
        sys.exit(0)
    if sys.argv[1:] == ["g9"]:
        g9_phase_generic(ref)
        sys.exit(0)
    if sys.argv[1:] == ["g8"]:
        g8_scfpyr_full(ref)
        sys.exit(0)
    g1_masks(ref)
    g2_pyramid(ref)
    g3_extract(ref)
    kats(ref)
    g4_head(ref)
    g7_sampler(ref)
    g8_scfpyr_full(ref)
    g9_phase_generic(ref)
    g10_train_phase(ref)
    g11_nondefault(ref)
    g12_sampler_keywords(ref)
    g13_resnet50_extractor_plumbing(ref)
    os.system("ls -la %s" % HERE)
