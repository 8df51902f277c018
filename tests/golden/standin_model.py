"""A STAND-IN for the third-party model file the reference loads by name.

TEST INFRASTRUCTURE ONLY.  The reference's ResNet50 lives in `pytorch-benchmarks/ferplus/resnet50_ferplus_dag.{py,pth}`
(albanie/pytorch-benchmarks, fetched by URL, not vendored: SURVEY.md 8c) -- `api/utils/model_utils.py:65-79` executes that `.py`
with importlib, calls the function named like the model with `weights_path=`, and `api/resnet50_extractor.py:36-41,74-83` uses the
returned module's `meta`, `_modules['pool5_7x7_s1']` (forward hook), `.to(device)`, `.eval()`.

`write_benchmark_dir` writes a definition file and a `.pth` in that layout -- THIS BUILD'S OWN restatement of the Caffe-style
ResNet-50 graph with the public layer names (generated below from `weights.resnet50_layers()`, not copied from anywhere), loaded
with the build's deterministic weights -- so that

  * `tests/golden/make_golden.py g13` can run the REAL `Resnet50_Extractor(benchmark_dir, model_name, 'pool5_7x7_s1').get_vec`
    on it (the last reference code of the hot path that had never executed here: importlib exec, the hook on the named layer,
    `relu(squeeze())`, `meta`), and
  * the product's drop-in class can be pointed at the very same directory (it reads the definition with `ast`, never executes it).

It pins plumbing, not arithmetic: the convolutions are this build's restatement either way (parity for ResNet50 stays unpinned).
"""
import os

import numpy as np
import torch


def definition_source(layers, stages, mean, eps=1e-5, ceil_mode=True):
    """Text of `<model_name>.py`: class Resnet50_ferplus_dag(nn.Module) with one attribute per layer (`conv2_1_1x1_reduce`,
    `conv2_1_1x1_reduce_bn`, ... `pool5_7x7_s1`, `classifier`), `self.meta`, and the factory `resnet50_ferplus_dag(weights_path)`."""
    init, fwd = [], []
    for name, cin, cout, k, s, p in layers:
        init.append("        self.%s = nn.Conv2d(%d, %d, kernel_size=[%d, %d], stride=(%d, %d), padding=(%d, %d), bias=False)"
                    % (name, cin, cout, k, k, s, s, p, p))
        init.append("        self.%s_bn = nn.BatchNorm2d(%d, eps=%r, momentum=0.1, affine=True, track_running_stats=True)" % (name, cout, eps))
        if name == "conv1_7x7_s2":
            init.append("        self.conv1_relu_7x7_s2 = nn.ReLU()")
            init.append("        self.pool1_3x3_s2 = nn.MaxPool2d(kernel_size=[3, 3], stride=[2, 2], padding=(0, 0), dilation=1, ceil_mode=%s)"
                        % bool(ceil_mode))
    init.append("        self.pool5_7x7_s1 = nn.AvgPool2d(kernel_size=[7, 7], stride=[1, 1], padding=0)")
    init.append("        self.classifier = nn.Conv2d(2048, 8, kernel_size=[1, 1], stride=(1, 1))")
    fwd.append("        x = self.pool1_3x3_s2(self.conv1_relu_7x7_s2(self.conv1_7x7_s2_bn(self.conv1_7x7_s2(data))))")
    for stage, blocks, _, _, _ in stages:
        for b in range(1, blocks + 1):
            pre = "conv%d_%d_" % (stage, b)
            if b == 1:
                fwd.append("        sc = self.%s1x1_proj_bn(self.%s1x1_proj(x))" % (pre, pre))
            else:
                fwd.append("        sc = x")
            fwd.append("        y = torch.relu(self.%s1x1_reduce_bn(self.%s1x1_reduce(x)))" % (pre, pre))
            fwd.append("        y = torch.relu(self.%s3x3_bn(self.%s3x3(y)))" % (pre, pre))
            fwd.append("        y = self.%s1x1_increase_bn(self.%s1x1_increase(y))" % (pre, pre))
            fwd.append("        x = torch.relu(y + sc)")
    fwd.append("        pool5 = self.pool5_7x7_s1(x)")
    fwd.append("        return self.classifier(pool5)")
    return "\n".join([
        "# stand-in written by tests/golden/standin_model.py (this build's restatement of the graph; NOT the third-party file)",
        "import torch",
        "import torch.nn as nn",
        "",
        "",
        "class Resnet50_ferplus_dag(nn.Module):",
        "",
        "    def __init__(self):",
        "        super(Resnet50_ferplus_dag, self).__init__()",
        "        self.meta = {'mean': [%s]," % ", ".join(repr(float(m)) for m in mean),
        "                     'std': [1, 1, 1],",
        "                     'imageSize': [224, 224, 3]}",
    ] + init + ["", "    def forward(self, data):"] + fwd + [
        "",
        "",
        "def resnet50_ferplus_dag(weights_path=None, **kwargs):",
        "    model = Resnet50_ferplus_dag()",
        "    if weights_path:",
        "        state_dict = torch.load(weights_path)",
        "        model.load_state_dict(state_dict)",
        "    return model",
        ""])


def write_benchmark_dir(root, weights_mod, seed=5, model_name="resnet50_ferplus_dag", mean=None, eps=1e-5, ceil_mode=True):
    """`<root>/ferplus/<model_name>.{py,pth}`; returns (benchmark_dir, numpy state_dict of the trunk)."""
    d = os.path.join(str(root), "ferplus")
    os.makedirs(d, exist_ok=True)
    mean = list(weights_mod.RESNET50_MEAN) if mean is None else list(mean)
    src = definition_source(weights_mod.resnet50_layers(), weights_mod.RESNET50_STAGES, mean, eps, ceil_mode)
    with open(os.path.join(d, model_name + ".py"), "w") as f:
        f.write(src.replace("Resnet50_ferplus_dag", model_name[0].upper() + model_name[1:]).replace("def resnet50_ferplus_dag", "def " + model_name))
    sd = weights_mod.make_resnet50_state_dict(seed=seed)
    on_disk = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    for k in list(on_disk):
        if k.endswith("_bn.running_var"):
            on_disk[k[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    on_disk["classifier.weight"] = torch.from_numpy(weights_mod.det_uniform("classifier.weight", (8, 2048, 1, 1), -0.02, 0.02, seed))
    on_disk["classifier.bias"] = torch.zeros(8)
    torch.save(on_disk, os.path.join(d, model_name + ".pth"))
    return str(root), sd
