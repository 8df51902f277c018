"""CPU oracle for the MIMAMO-Net per-video inference hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module, and only as the checker /
reported baseline.  The product package (mimamo-net_amd/) never imports it and fails
loudly when its HIP library is missing.

What it is: an independent restatement, on numpy + PyTorch-CPU ops, of the reference's
algorithm for the path named by BASELINE.json `north_star` (paths relative to
/root/reference):

  pyramid          api/steerable/SCFpyr_PyTorch.py:70-208, api/steerable/math_utils.py:25-73
  phase difference api/phase_difference_extractor.py:38-134, api/utils/phase_utils.py:5-40,78-129
  orchestration    api/tester.py:122-139, api/sampler/snippet_sampler.py:107-152, api/tester.py:94-121
  two-stream head  api/mimamo_net.py:6-143
  ResNet50 pool5   api/resnet50_extractor.py:74-83 (+ third-party albanie/pytorch-benchmarks
                   resnet50_ferplus_dag, NOT vendored in the reference and not on this filesystem)

Pinning status (see tests/test_oracle_golden.py, tests/golden/make_golden.py):
  * pyramid, masks, crop indices, phase difference, unwrap/diff/blur KATs, two-stream head
    (incl. the GRU seq-over-dim-0 quirk), sampler ranges: PINNED against outputs of the real
    reference imported in the build container (fixtures tests/golden/*.npz).
  * ResNet50 pool5: PARITY UNPINNED -- the reference delegates the arithmetic to a model
    file + weights fetched by URL with no pinned version (api/readme.md:60-74).  The layer
    graph below is the publicly documented Caffe-style ResNet-50 of that model file; only
    the hook layer name, 2048-d output, std==[1,1,1] -> x255 convention and the square
    224 input are confirmed by reference code (resnet50_extractor.py:15,77;
    utils/model_utils.py:27-28,37-38).

All functions take/return numpy arrays unless noted.  `dtype` selects the arithmetic
precision (np.float32 mirrors the reference; np.float64 is the noise-floor reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PI = math.pi


# --------------------------------------------------------------------------------------
# Steerable pyramid  (SCFpyr_PyTorch.py, math_utils.py)
# --------------------------------------------------------------------------------------
def prepare_grid(m, n):
    """math_utils.py:52-60.  Normalised frequency grid; centre sample patched."""
    x = np.linspace(-(m // 2) / (m / 2), (m // 2) / (m / 2) - (1 - m % 2) * 2 / m, num=m)
    y = np.linspace(-(n // 2) / (n / 2), (n // 2) / (n / 2) - (1 - n % 2) * 2 / n, num=n)
    xv, yv = np.meshgrid(y, x)
    angle = np.arctan2(yv, xv)
    rad = np.sqrt(xv ** 2 + yv ** 2)
    rad[m // 2][n // 2] = rad[m // 2][n // 2 - 1]
    return np.log2(rad), angle


def rcos_fn(width=1.0, position=-0.5):
    """math_utils.py:62-69.  259-point raised-cosine transition table."""
    n = 256
    x = np.pi * np.arange(-n - 1, 2) / 2 / n
    y = np.cos(x) ** 2
    y[0] = y[1]
    y[n + 2] = y[n + 1]
    x = position + 2 * width / np.pi * (x + np.pi / 4)
    return x, y


def point_op(im, y, x):
    """math_utils.py:71-73 (np.interp, clamped at the table ends)."""
    return np.interp(im.ravel(), x, y).reshape(im.shape)


def crop_bounds(d):
    """SCFpyr_PyTorch.py:182-183.  Integer, must be bit-exact: 96->[24,72) 48->[12,36)."""
    start = int(np.ceil((d + 0.5) / 2) - np.ceil((np.ceil((d - 0.5) / 2) + 0.5) / 2))
    end = int(start + np.ceil((d - 0.5) / 2))
    return start, end


def _factorial(n):
    return 1 if n <= 1 else n * _factorial(n - 1)


def angle_lut(nbands, lutsize=1024):
    """SCFpyr_PyTorch.py:61-63,148-150."""
    xcosn = np.pi * np.arange(-(2 * lutsize + 1), lutsize + 2) / lutsize
    alpha = (xcosn + np.pi) % (2 * np.pi) - np.pi
    order = nbands - 1
    const = np.power(2, 2 * order) * np.square(_factorial(order)) / (nbands * _factorial(2 * order))
    ycosn = 2 * np.sqrt(const) * np.power(np.cos(xcosn), order) * (np.abs(alpha) < np.pi / 2)
    return xcosn, ycosn


def pyramid_masks(size, height, nbands):
    """All real masks of SCFpyr_PyTorch.build/_build_levels for a square `size` input, float64.

    Returns dict: lo0, hi0, levels=[{himask, anglemask[b], lomask, crop=(s,e), size}] for the
    height-2 band-pass levels (list item l+1 of the pyramid is levels[l]).
    """
    log_rad, angle = prepare_grid(size, size)
    xr, yr = rcos_fn(1, -0.5)
    yr = np.sqrt(yr)
    yir = np.sqrt(1 - yr ** 2)
    out = {"lo0": point_op(log_rad, yir, xr), "hi0": point_op(log_rad, yr, xr), "levels": []}
    xcosn, ycosn = angle_lut(nbands)
    for _ in range(height - 2):  # _build_levels recursion with height-1 .. 2
        xr = xr - np.log2(2)
        lev = {"size": log_rad.shape[0], "himask": point_op(log_rad, yr, xr), "anglemask": []}
        for b in range(nbands):
            lev["anglemask"].append(point_op(angle, ycosn, xcosn + np.pi * b / nbands))
        s, e = crop_bounds(log_rad.shape[0])
        log_rad = log_rad[s:e, s:e]
        angle = angle[s:e, s:e]
        lev["crop"] = (s, e)
        lev["lomask"] = point_op(log_rad, np.abs(np.sqrt(1 - yr ** 2)), xr)
        out["levels"].append(lev)
    return out


def symmetric_extension(x):
    """phase_utils.py:116-129: [..., H, W] -> [..., 2H, 2W] = [[x, flipLR],[flipUD, flipBoth]]."""
    top = np.concatenate([x, x[..., :, ::-1]], axis=-1)
    return np.concatenate([top, top[..., ::-1, :]], axis=-2)


def _t(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.float32 if dtype == np.float32 else torch.float64)


def pyramid_build(images, height=4, nbands=2, dtype=np.float32, keep_residuals=False):
    """SCFpyr_PyTorch.build (70-125) for images [N, S, S] (already mirrored if wanted).

    Returns the band-pass levels only: list over levels of complex arrays [nbands, N, s, s]
    (level l has s = S / 2**l).  The hi-pass / low-pass residuals (build:120-124,
    _build_levels:129-135) are not used by inference; keep_residuals=True returns them too.
    """
    n, s, s2 = images.shape
    assert s == s2, "square input only (SCFpyr_PyTorch.py:87 swaps height/width)"
    if height > int(np.floor(np.log2(s)) - 2):
        raise RuntimeError("Cannot build {} levels, image too small.".format(height))
    if nbands < 2:
        raise ValueError("nbands < 2 unsupported (math_utils.py:79-84 recurses forever)")
    masks = pyramid_masks(s, height, nbands)
    x = _t(images, dtype)
    rdt = x.dtype
    dft = torch.fft.fftshift(torch.fft.fft2(x), dim=(-2, -1))
    lodft = dft * _t(masks["lo0"], np.float64).to(rdt)
    fact = complex(0, -1) ** (nbands - 1)
    levels = []
    for lev in masks["levels"]:
        himask = _t(lev["himask"], np.float64).to(rdt)
        bands = []
        for b in range(nbands):
            am = _t(lev["anglemask"][b], np.float64).to(rdt)
            banddft = lodft * am * himask
            re = fact.real * banddft.real - fact.imag * banddft.imag
            im = fact.real * banddft.imag + fact.imag * banddft.real
            banddft = torch.complex(re, im)
            bands.append(torch.fft.ifft2(torch.fft.ifftshift(banddft, dim=(-2, -1))))
        levels.append(torch.stack(bands, 0).numpy())
        s0, e0 = lev["crop"]
        lodft = lodft[:, s0:e0, s0:e0] * _t(lev["lomask"], np.float64).to(rdt)
    if keep_residuals:
        hi = torch.fft.ifft2(torch.fft.ifftshift(dft * _t(masks["hi0"], np.float64).to(rdt), dim=(-2, -1))).real
        lo = torch.fft.ifft2(torch.fft.ifftshift(lodft, dim=(-2, -1))).real
        return levels, hi.numpy(), lo.numpy()
    return levels


def build_pyramid(im_batch, height=4, nbands=2, extract_level=(1, 2), symmetry=True, dtype=np.float32):
    """Phase_Difference_Extractor.build_pyramid (phase_difference_extractor.py:38-87).

    im_batch [B, P, W, H] -> list (one per extract level) of float arrays [B, nbands, P, w, h, 2].
    """
    b, p, w, h = im_batch.shape
    ims = im_batch.reshape(b * p, w, h)
    if symmetry:
        ims = symmetric_extension(ims)
    levels = pyramid_build(ims, height, nbands, dtype)
    single = isinstance(extract_level, int)
    out = []
    for lv in ([extract_level] if single else list(extract_level)):
        c = levels[lv - 1]  # pyramid list item lv (item 0 is the hi-pass residual)
        nb, _, ww, hh = c.shape
        c = c.reshape(nb, b, p, ww, hh).transpose(1, 0, 2, 3, 4)
        if symmetry:
            c = c[..., : ww // 2, : hh // 2]
        out.append(np.ascontiguousarray(np.stack([c.real, c.imag], -1)))
    return out[0] if single else out


# --------------------------------------------------------------------------------------
# Phase difference  (phase_utils.py, phase_difference_extractor.py:93-134)
# --------------------------------------------------------------------------------------
def diff(x, axis):
    """phase_utils.py:21-40 (n=1)."""
    x = np.moveaxis(x, axis, 0)
    return np.moveaxis(x[1:] - x[:-1], 0, axis)


def unwrap(p, axis, dtype=None):
    """phase_utils.py:5-20.  C `fmod` semantics: only positive jumps get corrected (quirk Q2)."""
    p = np.moveaxis(np.asarray(p), axis, 0)
    dt = p.dtype if dtype is None else dtype
    p = p.astype(dt)
    pi = dt.type(PI) if hasattr(dt, "type") else np.dtype(dt).type(PI)
    two_pi = np.dtype(dt).type(2 * PI)
    dd = p[1:] - p[:-1]
    ddmod = np.fmod(dd + pi, two_pi) - pi
    ddmod = np.where((ddmod == -pi) & (dd > 0), pi, ddmod)
    corr = ddmod - dd
    corr = np.where(np.abs(dd) < pi, np.dtype(dt).type(0), corr)
    up = p.copy()
    # torch.cumsum on CPU accumulates sequentially in the tensor dtype
    acc = np.zeros_like(corr[0])
    for i in range(corr.shape[0]):
        acc = acc + corr[i]
        up[i + 1] = p[i + 1] + acc
    return np.moveaxis(up, 0, axis)


def gaussian_kernel(std=2, tap=11):
    """phase_utils.py:108-115 (float64, unnormalised; sum = 24.859243613...)."""
    r = np.arange(tap) - tap // 2
    return np.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / (2.0 * std ** 2))


def amplitude_blur(mag, phase, kernel):
    """phase_utils.py:78-90: depthwise zero-padded conv(mag*phase)/conv(mag); [B,C,W,H]."""
    tm = torch.from_numpy(np.ascontiguousarray(mag))
    tp = torch.from_numpy(np.ascontiguousarray(phase))
    c = tm.shape[1]
    k = torch.from_numpy(kernel).to(tm.dtype)[None, None].repeat(c, 1, 1, 1)
    pad = kernel.shape[0] // 2
    num = F.conv2d(tm * tp, k, groups=c, padding=pad)
    den = F.conv2d(tm, k, groups=c, padding=pad)
    return (num / den).numpy()


def extract(coeff, dtype=np.float32):
    """Phase_Difference_Extractor.extract (phase_difference_extractor.py:93-134).

    coeff [B, nbands, P, W, H, 2] -> [B, nbands, P-1, W, H].
    """
    coeff = np.asarray(coeff).astype(dtype)
    b, nb, p, w, h, _ = coeff.shape
    re = torch.from_numpy(np.ascontiguousarray(coeff[..., 0]))
    im = torch.from_numpy(np.ascontiguousarray(coeff[..., 1]))
    phase = torch.atan2(im, re).numpy().reshape(b * nb, p, w, h)
    mag = (torch.sqrt(im * im + re * re).numpy().reshape(b * nb, p, w, h) + np.dtype(dtype).type(1e-10))
    mag = mag.astype(dtype)
    phase = unwrap(phase, axis=1)
    den = amplitude_blur(mag, phase, gaussian_kernel(2, 11)).reshape(b, nb, p, w, h)
    d = diff(den, axis=2)
    d = d - d.mean(-1).mean(-1)[..., None, None]
    lim = np.dtype(dtype).type(5 * PI)
    return np.clip(d, -lim, lim).astype(dtype)


def extract_phase(coeff, return_phase=False, return_both=False, dtype=np.float32):
    """Training-side Steerable_Pyramid_Phase.extract_phase (Aff-wild-exps/utils.py:367-418): extract() plus the options
    to return the mean-centred denoised phase (:410-412,417) or insert_tensors(diff, denoised[1:]) (:413-416,419-432 --
    whose loop runs over half of the result, the other half stays zero).  Pinned by G10 (tests/golden/train_phase.npz): the
    real class run on the float64 cast of its fp32 coefficients (its blur casts the kernel to float32 only on CUDA,
    utils.py:254, so float32 data raises on this CPU-only host while float64 data runs)."""
    coeff = np.asarray(coeff).astype(dtype)
    b, nb, p, w, h, _ = coeff.shape
    re = torch.from_numpy(np.ascontiguousarray(coeff[..., 0]))
    im = torch.from_numpy(np.ascontiguousarray(coeff[..., 1]))
    phase = torch.atan2(im, re).numpy().reshape(b * nb, p, w, h)
    mag = (torch.sqrt(im * im + re * re).numpy().reshape(b * nb, p, w, h) + np.dtype(dtype).type(1e-10)).astype(dtype)
    den = amplitude_blur(mag, unwrap(phase, axis=1), gaussian_kernel(2, 11)).reshape(b, nb, p, w, h)
    d = diff(den, axis=2)
    den_c = den - den.mean(-1).mean(-1)[..., None, None]
    d = d - d.mean(-1).mean(-1)[..., None, None]
    lim = np.dtype(dtype).type(5 * PI)
    d = np.clip(d, -lim, lim).astype(dtype)
    if return_both:
        t_a, t_b = d, den_c[:, :, 1:]
        length = t_a.shape[2]
        res = np.zeros((b, nb, 2 * length, w, h), dtype=dtype)
        for i in range(length):
            res[:, :, i] = (t_a if i % 2 == 0 else t_b)[:, :, i // 2]
        return res
    return den_c.astype(dtype) if return_phase else d


def phase_diff_output(phase_batch, height=4, nbands=2, extract_level=(1, 2), dtype=np.float32):
    """Tester.phase_diff_output (tester.py:122-139): [bs,T,P,W,H] -> (phase_0, phase_1)."""
    bs, t, p, w, h = phase_batch.shape
    coeffs = build_pyramid(phase_batch.reshape(bs * t, p, w, h), height, nbands, extract_level, True, dtype)
    outs = []
    for c in coeffs:
        d = extract(c, dtype)
        n, nb, k, ww, hh = d.shape
        outs.append(d.reshape(bs, t, nb * k, ww, hh))
    return tuple(outs)


def phase_diff_from_frames(frames, window_ids, **kw):
    """Deduplicated driver: unique frames [N,W,H] + window ids [J,P] -> ([J,24,48,48],[J,24,24,24]).

    Same arithmetic as phase_diff_output on the gathered windows (pyramid is per-frame, so
    building it once per unique frame is exact -- SURVEY.md quirk Q3).
    """
    dtype = kw.get("dtype", np.float32)
    n, w, h = frames.shape
    coeffs = build_pyramid(frames.reshape(n, 1, w, h), kw.get("height", 4), kw.get("nbands", 2),
                           kw.get("extract_level", (1, 2)), True, dtype)
    outs = []
    ids = np.asarray(window_ids)
    for c in coeffs:  # [N, nb, 1, w, h, 2]
        g = c[:, :, 0][ids]  # [J, P, nb, w, h, 2]
        g = np.ascontiguousarray(g.transpose(0, 2, 1, 3, 4, 5))
        d = extract(g, dtype)
        j, nb, k, ww, hh = d.shape
        outs.append(d.reshape(j, nb * k, ww, hh))
    return tuple(outs)


# --------------------------------------------------------------------------------------
# Sampler / assembly index logic (snippet_sampler.py:107-152, tester.py:94-121)
# --------------------------------------------------------------------------------------
def snippet_ranges(n_frames, length=64, stride=64):
    """snippet_sampler.py:112-126 incl. the short-video rule and the tail snippet."""
    if n_frames < length:
        length = stride = n_frames
    ranges = []
    start, end = 0, length
    while end <= n_frames and start < n_frames:
        ranges.append([start, end])
        start += stride
        end = start + length
    assert len(ranges) != 0, "No snippet is sampled."
    if ranges[-1][1] < n_frames:
        ranges.append([n_frames - length, n_frames])
    return ranges


def window_ids(start, end, n_frames, num_phase=12):
    """snippet_sampler.py:144-152: clamped ids f + (i - num_phase//2), i = 0..num_phase."""
    out = np.empty((end - start, num_phase + 1), dtype=np.int64)
    for j, f in enumerate(range(start, end)):
        for i in range(num_phase + 1):
            out[j, i] = min(max(0, f + i - num_phase // 2), n_frames - 1)
    return out


def assemble(preds, ranges, n_labels=2):
    """tester.py:103-118: later snippets overwrite earlier ones; asserts full coverage."""
    max_len = max(r[1] for r in ranges)
    video = np.zeros((max_len, n_labels))
    lo, hi = 0, 0
    for (s, e), p in zip(ranges, preds):
        video[s:e, :] = p
        lo, hi = min(lo, s), max(hi, e)
    assert lo == 0 and hi == max_len
    return video


# --------------------------------------------------------------------------------------
# Two-stream head (mimamo_net.py)
# --------------------------------------------------------------------------------------
def _bn(x, sd, prefix, eps=1e-5):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, eps)


def _to_torch_sd(sd, dtype=torch.float32):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dtype) if isinstance(v, np.ndarray) else v.to(dtype))
            for k, v in sd.items() if "num_batches_tracked" not in k}


def mlp_forward(sd, rgb):
    """MLP(hidden_units) (mimamo_net.py:6-26), eval mode: [Dropout, Linear, BatchNorm1d, ReLU] per hidden layer, i.e. Linear at
    index 4i+1 and BN at 4i+2 of the Sequential.  rgb [n, hidden_units[0]] -> [n, 256]."""
    x, i = rgb, 0
    while ("mlp.mlp.%d.weight" % (4 * i + 1)) in sd:
        x = F.linear(x, sd["mlp.mlp.%d.weight" % (4 * i + 1)], sd["mlp.mlp.%d.bias" % (4 * i + 1)])
        x = F.relu(_bn(x, sd, "mlp.mlp.%d" % (4 * i + 2)))
        i += 1
    return x


def phasenet_forward(sd, p0, p1):
    """PhaseNet(48, 24, feature=True) (mimamo_net.py:27-95), eval mode. -> [n,256]."""
    def block(x, i):
        pre = "phasenet.conv_net.%d." % i
        x = F.conv2d(x, sd[pre + "0.weight"], sd[pre + "0.bias"], padding=1)
        x = F.relu(_bn(x, sd, pre + "1"))
        x = F.conv2d(x, sd[pre + "3.weight"], sd[pre + "3.bias"], stride=2, padding=1)
        return F.relu(_bn(x, sd, pre + "4"))
    x = block(p0, 0)
    x = torch.cat([x, p1], dim=1)
    x = block(x, 1)
    x = block(x, 2)
    x = F.avg_pool2d(x, 6).flatten(1)
    x = F.relu(F.linear(x, sd["phasenet.fc.0.weight"], sd["phasenet.fc.0.bias"]))
    x = _bn(x, sd, "phasenet.fc.2")
    x = F.relu(F.linear(x, sd["phasenet.fc.4.weight"], sd["phasenet.fc.4.bias"]))
    return _bn(x, sd, "phasenet.fc.6")


def gru_forward(sd, x, hidden=128, layers=2):
    """nn.GRU(256,128,bidirectional,num_layers=2) WITHOUT batch_first (mimamo_net.py:119,139).

    x [S, Bt, 256]: recurrence runs over dim 0 (= snippets, quirk Q1), dim 1 is the GRU batch.
    Gate order r,z,n; n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h.
    """
    s, bt, _ = x.shape
    inp = x
    for l in range(layers):
        outs = []
        for suffix, order in (("", range(s)), ("_reverse", range(s - 1, -1, -1))):
            wih, whh = sd["rnns.weight_ih_l%d%s" % (l, suffix)], sd["rnns.weight_hh_l%d%s" % (l, suffix)]
            bih, bhh = sd["rnns.bias_ih_l%d%s" % (l, suffix)], sd["rnns.bias_hh_l%d%s" % (l, suffix)]
            h = torch.zeros(bt, hidden, dtype=x.dtype)
            seq = [None] * s
            for t in order:
                gi = F.linear(inp[t], wih, bih)
                gh = F.linear(h, whh, bhh)
                r = torch.sigmoid(gi[:, :hidden] + gh[:, :hidden])
                z = torch.sigmoid(gi[:, hidden:2 * hidden] + gh[:, hidden:2 * hidden])
                n = torch.tanh(gi[:, 2 * hidden:] + r * gh[:, 2 * hidden:])
                h = (1 - z) * n + z * h
                seq[t] = h
            outs.append(torch.stack(seq, 0))
        inp = torch.cat(outs, dim=-1)
    return inp


def two_stream_forward(state_dict, phase_0, phase_1, rgb, dtype=np.float32):
    """Two_Stream_RNN.forward (mimamo_net.py:129-143), eval mode.

    phase_0 [bs,T,24,48,48], phase_1 [bs,T,24,24,24], rgb [bs,T,2048] -> [bs,T,2].
    """
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    sd = _to_torch_sd(state_dict, tdt)
    p0 = torch.from_numpy(np.ascontiguousarray(phase_0)).to(tdt)
    p1 = torch.from_numpy(np.ascontiguousarray(phase_1)).to(tdt)
    r = torch.from_numpy(np.ascontiguousarray(rgb)).to(tdt)
    bs, t = r.shape[0], r.shape[1]
    with torch.no_grad():
        spatial = mlp_forward(sd, r.reshape(bs * t, -1))
        temporal = phasenet_forward(sd, p0.reshape((bs * t,) + tuple(p0.shape[2:])),
                                    p1.reshape((bs * t,) + tuple(p1.shape[2:])))
        f = torch.cat([spatial, temporal], dim=-1)
        f = F.relu(F.linear(f, sd["transform.0.weight"], sd["transform.0.bias"]))
        f = _bn(f, sd, "transform.2")
        o = gru_forward(sd, f.reshape(bs, t, -1))
        o = F.linear(o.reshape(bs * t, -1), sd["classifier.1.weight"], sd["classifier.1.bias"])
        o = _bn(o, sd, "classifier.2")
    return o.reshape(bs, t, -1).numpy()


# --------------------------------------------------------------------------------------
# ResNet50 pool5 (third-party albanie resnet50_ferplus_dag; PARITY UNPINNED, see header)
# --------------------------------------------------------------------------------------
RESNET50_STAGES = ((2, 3, 64, 256, 1), (3, 4, 128, 512, 2), (4, 6, 256, 1024, 2), (5, 3, 512, 2048, 2))
RESNET50_MEAN = (131.0912, 103.8827, 91.4953)


def resnet50_layer_list(stride_on_first_1x1=True):
    """[(name, cin, cout, k, stride, pad)] in forward order, Caffe-style ResNet-50 trunk."""
    layers = [("conv1_7x7_s2", 3, 64, 7, 2, 3)]
    cin = 64
    for stage, blocks, mid, cout, stride in RESNET50_STAGES:
        for b in range(1, blocks + 1):
            s = stride if b == 1 else 1
            s1, s3 = (s, 1) if stride_on_first_1x1 else (1, s)
            pre = "conv%d_%d_" % (stage, b)
            if b == 1:
                layers.append((pre + "1x1_proj", cin, cout, 1, s, 0))
            layers.append((pre + "1x1_reduce", cin, mid, 1, s1, 0))
            layers.append((pre + "3x3", mid, mid, 3, s3, 1))
            layers.append((pre + "1x1_increase", mid, cout, 1, 1, 0))
            cin = cout
    return layers


def resnet50_pool5(state_dict, x, stride_on_first_1x1=True, ceil_mode=True, eps=1e-5, dtype=np.float32, channels_last=False):
    """Forward to `pool5_7x7_s1` and relu(squeeze) (resnet50_extractor.py:74-83).

    x [B,3,224,224] (already 255*x - mean) -> [B,2048].  Conv -> BN(eval) -> ReLU, bottleneck
    residual add then ReLU, MaxPool 3x3 s2 pad 0 ceil_mode, AvgPool 7x7.
    channels_last: the same ops on torch.channels_last tensors (bench.py's "as tuned" CPU figure: on AMD hosts oneDNN's NCHW 1x1
    convolution -- the reference's layout -- runs two orders of magnitude below its channels-last path); the reference itself
    uses the default NCHW format (False).
    """
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    sd = _to_torch_sd(state_dict, tdt)
    x = torch.from_numpy(np.ascontiguousarray(x)).to(tdt)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
        sd = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()}

    def cbr(x, name, stride, pad, relu=True):
        y = F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=pad)
        y = _bn(y, sd, name + "_bn", eps)
        return F.relu(y) if relu else y

    with torch.no_grad():
        x = cbr(x, "conv1_7x7_s2", 2, 3)
        x = F.max_pool2d(x, 3, 2, 0, ceil_mode=ceil_mode)
        for stage, blocks, mid, cout, stride in RESNET50_STAGES:
            for b in range(1, blocks + 1):
                s = stride if b == 1 else 1
                s1, s3 = (s, 1) if stride_on_first_1x1 else (1, s)
                pre = "conv%d_%d_" % (stage, b)
                sc = cbr(x, pre + "1x1_proj", s, 0, relu=False) if b == 1 else x
                y = cbr(x, pre + "1x1_reduce", s1, 0)
                y = cbr(y, pre + "3x3", s3, 1)
                y = cbr(y, pre + "1x1_increase", 1, 0, relu=False)
                x = F.relu(y + sc)
        x = F.avg_pool2d(x, 7, 1)
        x = F.relu(x.reshape(x.shape[0], -1))
    return x.numpy()
