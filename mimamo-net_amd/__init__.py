"""mimamo-net_amd: MI355X-native (gfx950) implementation of MIMAMO-Net's per-video inference
hot path (steerable pyramid + phase difference -> ResNet50 pool5 -> two-stream GRU head).

The directory name carries a hyphen (it mirrors the reference repo name); import it either
with ``importlib.import_module("mimamo-net_amd")`` or through the root-level alias module
``mimamo_net_amd``.  Heavy submodules (those that load libmimamo_hip.so) are imported lazily.
"""
__version__ = "0.1.0"

_LAZY = {
    "Phase_Difference_Extractor": ".phase_difference_extractor",
    "SCFpyr_PyTorch": ".scfpyr",
    "Steerable_Pyramid_Phase": ".phase_difference_extractor",
    "Resnet50_Extractor": ".resnet50_extractor",
    "Two_Stream_RNN": ".mimamo_net",
    "Tester": ".tester",
    "Snippet_Sampler": ".sampler",
    "Image_Sampler": ".sampler",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module(_LAZY[name], __name__)
        return getattr(mod, name)
    raise AttributeError(name)
