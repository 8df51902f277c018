"""Deterministic parameter generators + state_dict layouts for the hot path.

There is no network in the build/bench environment, so the published checkpoints
(api/models/download_models.sh:1, api/readme.md:60-74) cannot be fetched.  The bench and
the parity tests use random-init weights of the reference architecture produced by a
closed-form, platform-independent generator (splitmix64 over (tensor name, flat index)) --
NOT torch.manual_seed, whose stream differs across versions/devices.

Key layouts:
  two-stream head : the 107-tensor state_dict of Two_Stream_RNN (api/mimamo_net.py:96-122)
  ResNet50        : `<conv>.weight` [+ `.bias`] and `<conv>_bn.{weight,bias,running_mean,
                    running_var}` with the Caffe-style layer names of the third-party
                    resnet50_ferplus_dag model file (conv1_7x7_s2, conv2_1_1x1_reduce, ...).
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name):
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def det_uniform(name, shape, lo=-1.0, hi=1.0, seed=0):
    """float32 array, value(i) = lo + (hi-lo) * u(i), u from splitmix64(fnv(name)^seed + i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + base) & _MASK
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits -> exact in float32
    u = bits.astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def _he(name, shape, fan_in, seed, gain=1.0):
    b = gain * float(np.sqrt(6.0 / fan_in))
    return det_uniform(name, shape, -b, b, seed)


def _bn(sd, prefix, c, seed, gamma=(0.8, 1.2)):
    sd[prefix + ".weight"] = det_uniform(prefix + ".weight", (c,), gamma[0], gamma[1], seed)
    sd[prefix + ".bias"] = det_uniform(prefix + ".bias", (c,), -0.1, 0.1, seed)
    sd[prefix + ".running_mean"] = det_uniform(prefix + ".running_mean", (c,), -0.1, 0.1, seed)
    sd[prefix + ".running_var"] = det_uniform(prefix + ".running_var", (c,), 0.8, 1.25, seed)


def _lin(sd, prefix, cout, cin, seed, gain=1.0):
    sd[prefix + ".weight"] = _he(prefix + ".weight", (cout, cin), cin, seed, gain)
    sd[prefix + ".bias"] = det_uniform(prefix + ".bias", (cout,), -0.05, 0.05, seed)


def _conv(sd, prefix, cout, cin, k, seed, bias=True, gain=1.0):
    sd[prefix + ".weight"] = _he(prefix + ".weight", (cout, cin, k, k), cin * k * k, seed, gain)
    if bias:
        sd[prefix + ".bias"] = det_uniform(prefix + ".bias", (cout,), -0.05, 0.05, seed)


MLP_PUBLISHED = (2048, 256, 256)   # api/tester.py:45


def two_stream_bn_keys(mlp_units=MLP_PUBLISHED):
    """BatchNorm modules of Two_Stream_RNN in module order (each also owns an int64 `num_batches_tracked`)."""
    return tuple("mlp.mlp.%d" % (4 * i + 2) for i in range(len(mlp_units) - 1)) + TWO_STREAM_BN_KEYS[2:]


TWO_STREAM_BN_KEYS = ("mlp.mlp.2", "mlp.mlp.6", "phasenet.conv_net.0.1", "phasenet.conv_net.0.4",
                      "phasenet.conv_net.1.1", "phasenet.conv_net.1.4", "phasenet.conv_net.2.1",
                      "phasenet.conv_net.2.4", "phasenet.fc.2", "phasenet.fc.6", "phasenet.classifier.1",
                      "transform.2", "classifier.2")


def make_two_stream_state_dict(seed=0, num_phase=12, n_out=2, mlp_units=MLP_PUBLISHED):
    """Random-init state_dict with the exact key/shape layout of Two_Stream_RNN (107 tensors
    incl. the 13 `num_batches_tracked` counters for the published configuration).  n_out = len(label_name.split('_'));
    mlp_units = mlp_hidden_units (api/mimamo_net.py:6-26: Linear at Sequential index 4i+1, BatchNorm1d at 4i+2)."""
    sd = {}
    for i in range(len(mlp_units) - 1):
        _lin(sd, "mlp.mlp.%d" % (4 * i + 1), mlp_units[i + 1], mlp_units[i], seed)
        _bn(sd, "mlp.mlp.%d" % (4 * i + 2), mlp_units[i + 1], seed)
    nch = 2 * num_phase
    chans = [(nch, 64), (nch + 64, 128), (128, 256)]
    for i, (cin, cout) in enumerate(chans):
        pre = "phasenet.conv_net.%d." % i
        _conv(sd, pre + "0", cout, cin, 3, seed)
        _bn(sd, pre + "1", cout, seed)
        _conv(sd, pre + "3", cout, cout, 3, seed)
        _bn(sd, pre + "4", cout, seed)
    _lin(sd, "phasenet.fc.0", 256, 256, seed)
    _bn(sd, "phasenet.fc.2", 256, seed)
    _lin(sd, "phasenet.fc.4", 256, 256, seed)
    _bn(sd, "phasenet.fc.6", 256, seed)
    _lin(sd, "phasenet.classifier.0", 1, 256, seed)
    _bn(sd, "phasenet.classifier.1", 1, seed)
    _lin(sd, "transform.0", 256, 512, seed)
    _bn(sd, "transform.2", 256, seed)
    k = 1.0 / np.sqrt(128.0)
    for l in range(2):
        for sfx in ("", "_reverse"):
            for nm, shape in (("weight_ih", (384, 256)), ("weight_hh", (384, 128)),
                              ("bias_ih", (384,)), ("bias_hh", (384,))):
                key = "rnns.%s_l%d%s" % (nm, l, sfx)
                sd[key] = det_uniform(key, shape, -k, k, seed)
    _lin(sd, "classifier.1", n_out, 256, seed)
    _bn(sd, "classifier.2", n_out, seed)
    for bn in two_stream_bn_keys(mlp_units):
        sd[bn + ".num_batches_tracked"] = np.zeros((), dtype=np.int64)
    return sd


# (stage, blocks, mid, out, stride) -- Caffe-style ResNet-50
RESNET50_STAGES = ((2, 3, 64, 256, 1), (3, 4, 128, 512, 2), (4, 6, 256, 1024, 2), (5, 3, 512, 2048, 2))
# meta['mean'] of the third-party model (VGGFace2 statistics), std == [1,1,1]
RESNET50_MEAN = (131.0912, 103.8827, 91.4953)


def resnet50_layers(stride_on_first_1x1=True):
    """[(name, cin, cout, k, stride, pad)] in forward order."""
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


def make_resnet50_state_dict(seed=0, gamma_mid=(0.8, 1.2), gamma_out=(0.35, 0.55)):
    """Random-init ResNet-50 trunk (He-scaled convs; BN gamma~1, beta~0, var~1; the last BN
    of every bottleneck is damped so 16 residual adds keep activations O(1..10)).
    gamma_mid / gamma_out: BN weight ranges of the reduce + 3x3 layers / of the increase + projection layers; the
    numerical stress tests widen them (per-channel dynamic range, activations of 1e3 and beyond)."""
    sd = {}
    for name, cin, cout, k, _, _ in resnet50_layers():
        _conv(sd, name, cout, cin, k, seed, bias=False)
        out = name.endswith("1x1_increase") or name.endswith("1x1_proj")
        _bn(sd, name + "_bn", cout, seed, gamma=gamma_out if out else gamma_mid)
    return sd


# ---------------------------------------------------------------------------------------------
# Weight blobs handed to the C ABI (mm_resnet50_create / mm_head_create): flat float32 arrays,
# tensors concatenated in a fixed order.
# ---------------------------------------------------------------------------------------------
def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32).ravel()


def resnet50_blob(state_dict):
    """Per layer of resnet50_layers(): conv weight (OIHW), then BN weight, bias, running_mean, running_var.

    A conv bias b (the published model file declares `bias=False` everywhere, but a re-exported checkpoint may carry one) is
    folded into the BatchNorm that follows: BN(conv + b) = gamma * (conv - (mean - b)) / sqrt(var + eps) + beta, i.e. the blob's
    running_mean is mean - b (float64, rounded once)."""
    parts = []
    for name, cin, cout, k, _, _ in resnet50_layers():
        w = _np(state_dict[name + ".weight"])
        assert w.size == cout * cin * k * k, name
        parts.append(w)
        for s in ("weight", "bias", "running_mean", "running_var"):
            v = _np(state_dict[name + "_bn." + s])
            assert v.size == cout, name + "_bn." + s
            if s == "running_mean" and state_dict.get(name + ".bias") is not None:
                b = _np(state_dict[name + ".bias"])
                assert b.size == cout, name + ".bias"
                v = (v.astype(np.float64) - b.astype(np.float64)).astype(np.float32)
            parts.append(v)
    return np.concatenate(parts)


def read_model_definition(py_path):
    """What the third-party model DEFINITION file next to the weights says about the graph, without executing it.

    The reference executes `<benchmark_dir>/ferplus/<model_name>.py` and instantiates its class (api/utils/model_utils.py:65-79),
    then reads `model.meta` (api/resnet50_extractor.py:38-41).  Those generated files are flat lists of
    `self.<layer> = nn.<Type>(...)` assignments plus `self.meta = {...}`; this reads them with `ast` (literals only):
      meta                 {'mean', 'std', 'imageSize'} when the file assigns a literal dict to `self.meta`
      stride_on_first_1x1  True when `conv3_1_1x1_reduce` has stride 2 (Caffe style), False when `conv3_1_3x3` has it
      ceil_mode            `pool1_3x3_s2`'s ceil_mode
      bn_eps               eps of `conv1_7x7_s2_bn`
    Keys the file does not settle are absent from the result."""
    import ast
    with open(py_path, "r") as f:
        tree = ast.parse(f.read(), filename=py_path)

    def lit(node):
        try:
            return ast.literal_eval(node)
        except Exception:
            return None

    layers, out = {}, {}
    for node in ast.walk(tree):
        if not isinstance(node, ast.Assign) or len(node.targets) != 1:
            continue
        t = node.targets[0]
        if not (isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self"):
            continue
        if t.attr == "meta":
            m = lit(node.value)
            if isinstance(m, dict) and "mean" in m:
                out["meta"] = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in m.items()}
        elif isinstance(node.value, ast.Call):
            fn = node.value.func
            kind = fn.attr if isinstance(fn, ast.Attribute) else getattr(fn, "id", None)
            kw = {k.arg: lit(k.value) for k in node.value.keywords if k.arg}
            layers[t.attr] = (kind, [lit(a) for a in node.value.args], kw)

    def stride_of(name):
        if name not in layers or layers[name][0] != "Conv2d":
            return None
        _, args, kw = layers[name]
        s = kw.get("stride", args[3] if len(args) > 3 else 1)
        if isinstance(s, (list, tuple)):
            s = s[0]
        return int(s) if s is not None else None

    s1, s3 = stride_of("conv3_1_1x1_reduce"), stride_of("conv3_1_3x3")
    if s1 is not None and s3 is not None and {s1, s3} == {1, 2}:
        out["stride_on_first_1x1"] = s1 == 2
    if layers.get("pool1_3x3_s2", (None,))[0] == "MaxPool2d" and layers["pool1_3x3_s2"][2].get("ceil_mode") is not None:
        out["ceil_mode"] = bool(layers["pool1_3x3_s2"][2]["ceil_mode"])
    if layers.get("conv1_7x7_s2_bn", (None,))[0] == "BatchNorm2d":
        _, args, kw = layers["conv1_7x7_s2_bn"]
        eps = kw.get("eps", args[1] if len(args) > 1 else None)
        if eps is not None:
            out["bn_eps"] = float(eps)
    return out


_FLOAT_KEYS = {}


def two_stream_float_keys(mlp_units=MLP_PUBLISHED):
    """state_dict keys of Two_Stream_RNN in module order, without the int64 BN counters (the key set does not depend on
    num_phase, only two tensor shapes do)."""
    mlp_units = tuple(int(u) for u in mlp_units)
    if mlp_units not in _FLOAT_KEYS:
        _FLOAT_KEYS[mlp_units] = [k for k in make_two_stream_state_dict(0, mlp_units=mlp_units) if not k.endswith("num_batches_tracked")]
    return _FLOAT_KEYS[mlp_units]


_SHAPES = {}


def two_stream_shapes(mlp_units=MLP_PUBLISHED, num_phase=12, n_out=2):
    """{key: shape} of the float tensors for a configuration (what load_state_dict checks a checkpoint against)."""
    key = (tuple(int(u) for u in mlp_units), int(num_phase), int(n_out))
    if key not in _SHAPES:
        _SHAPES[key] = {k: tuple(np.shape(v)) for k, v in make_two_stream_state_dict(0, key[1], key[2], key[0]).items()
                        if not k.endswith("num_batches_tracked")}
    return _SHAPES[key]


def widen_classifier(state_dict):
    """A one-output head (label_name 'arousal' or 'valence': Linear(256,1) + BatchNorm1d(1), api/mimamo_net.py:120-122)
    in the library's two-output layout: output 0 is the checkpoint's, output 1 an inert row (zero weights, identity BN)
    that the caller slices away.  Output channels of the classifier GEMM are independent, so column 0 is unchanged."""
    sd = dict(state_dict)
    z = lambda v, fill: np.concatenate([_np(v).reshape((1,) + tuple(np.shape(v))[1:]),                     # noqa: E731
                                        np.full((1,) + tuple(np.shape(v))[1:], fill, dtype=np.float32)])
    sd["classifier.1.weight"] = z(state_dict["classifier.1.weight"], 0.0)
    sd["classifier.1.bias"] = z(state_dict["classifier.1.bias"], 0.0)
    sd["classifier.2.weight"] = z(state_dict["classifier.2.weight"], 1.0)
    sd["classifier.2.bias"] = z(state_dict["classifier.2.bias"], 0.0)
    sd["classifier.2.running_mean"] = z(state_dict["classifier.2.running_mean"], 0.0)
    sd["classifier.2.running_var"] = z(state_dict["classifier.2.running_var"], 1.0)
    return sd


def two_stream_blob(state_dict, mlp_units=MLP_PUBLISHED):
    return np.concatenate([_np(state_dict[k]) for k in two_stream_float_keys(mlp_units)])
