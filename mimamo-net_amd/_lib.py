"""ctypes binding of libmimamo_hip.so (include/mimamo_hip.h).  No fallbacks: if the library is
missing or a call fails, raise -- the product path never silently runs anything else."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MM_LIB_PATH: another build of the same library (tools/ A/B variants and -DMM_MEASURE builds under tools/_ab/, so that no measurement script
# ever copies a variant over the shipped file); unset everywhere else.  The loader says so on stderr -- never silently
LIB_PATH = os.environ.get("MM_LIB_PATH") or os.path.join(HERE, "libmimamo_hip.so")

MM_OK = 0
MM_ERR_INVALID_ARG = -1
MM_ERR_TOO_SMALL = -2
MM_ERR_UNSUPPORTED = -3
MM_ERR_HIP = -4
MM_ERR_NO_DEVICE = -5
MM_ERR_WORKSPACE = -6

_c = ctypes
_vp, _i, _i64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# name -> (restype, argtypes); must list EVERY symbol include/mimamo_hip.h declares
SIGNATURES = {
    "mm_version": (_i, []),
    "mm_status_string": (_c.c_char_p, [_i]),
    "mm_last_hip_error": (_i, []),
    "mm_pyramid_create": (_i, [_c.POINTER(_vp), _i, _i, _i, _i]),
    "mm_pyramid_destroy": (_i, [_vp]),
    "mm_pyramid_host_mask": (_i, [_i, _i, _i, _i, _i, _c.POINTER(_c.c_double), _c.POINTER(_i)]),
    "mm_pyramid_host_tables": (_i64, [_i, _i, _i, _i, _c.POINTER(_f), _i64]),
    "mm_pyramid_build": (_i, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp]),
    "mm_pyramid_build_batch": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "mm_phase_extract": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i, _i, _vp, _i, _i, _i, _vp]),
    "mm_phase_extract_generic": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "mm_phase_extract_generic_workspace_bytes": (_i64, [_i64, _i, _i, _i]),
    "mm_phase_extract_generic_ws": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    "mm_phase_workspace_bytes": (_i64, [_vp, _i64]),
    "mm_phase_diff_frames": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "mm_phase_diff_planes": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _i, _vp]),
    "mm_scfpyr_create": (_i, [_c.POINTER(_vp), _i, _i, _i, _i]),
    "mm_scfpyr_destroy": (_i, [_vp]),
    "mm_scfpyr_host_table": (_i, [_i, _i, _i, _i, _i, _c.POINTER(_c.c_double), _c.POINTER(_i), _c.POINTER(_i)]),
    "mm_scfpyr_num_outputs": (_i, [_vp]),
    "mm_scfpyr_output_info": (_i, [_vp, _i, _c.POINTER(_i), _c.POINTER(_i)]),
    "mm_scfpyr_workspace_bytes": (_i64, [_vp, _i64]),
    "mm_scfpyr_build": (_i, [_vp, _vp, _i, _i64, _c.POINTER(_vp), _vp, _i64, _vp]),
    "mm_preproc_host_coeffs": (_i, [_i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i]),
    "mm_preproc_create": (_i, [_c.POINTER(_vp), _i, _i, _i, _i, _c.POINTER(_f)]),
    "mm_preproc_destroy": (_i, [_vp]),
    "mm_preproc_forward": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "mm_profile_begin": (_i, []),
    "mm_profile_end": (_i, [_c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64)]),
    "mm_conv2d_nhwc": (_i, [_vp] * 7 + [_i] * 17 + [_vp]),
    "mm_resnet50_blob_floats": (_i64, []),
    "mm_resnet50_create": (_i, [_c.POINTER(_vp), _vp, _i64, _i, _i, _f]),
    "mm_resnet50_destroy": (_i, [_vp]),
    "mm_resnet50_set_winograd": (_i, [_vp, _i]),
    "mm_resnet50_set_precision": (_i, [_vp, _i]),
    "mm_resnet50_workspace_bytes": (_i64, [_vp, _i64]),
    "mm_resnet50_forward": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _i64, _vp]),
    "mm_head_blob_floats": (_i64, []),
    "mm_head_create": (_i, [_c.POINTER(_vp), _vp, _i64]),
    "mm_head_blob_floats_mlp": (_i64, [_i, _c.POINTER(_i)]),
    "mm_head_create_mlp": (_i, [_c.POINTER(_vp), _vp, _i64, _i, _c.POINTER(_i)]),
    "mm_head_blob_floats_cfg": (_i64, [_i, _c.POINTER(_i), _i]),
    "mm_head_create_cfg": (_i, [_c.POINTER(_vp), _vp, _i64, _i, _c.POINTER(_i), _i]),
    "mm_head_destroy": (_i, [_vp]),
    "mm_head_workspace_bytes": (_i64, [_vp, _i64, _i64]),
    "mm_head_forward": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
}

_lib = None


class MimamoHipError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().mm_status_string(status).decode()
        if status == MM_ERR_HIP:
            msg += " (hipError_t %d)" % lib().mm_last_hip_error()
        RuntimeError.__init__(self, "%s: %s" % (where, msg))


def lib():
    """Load (once) and return the ctypes library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmimamo_hip.so is missing at %s -- build it with `python -c \"import __graft_entry__ as g; "
                "g.build()\"` (hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7; import it FIRST so that this library binds to the same
        # HIP runtime instance (same SONAME -> the loader reuses it).  Loading ours first would pull in
        # /opt/rocm's runtime and leave torch on a mixed stack ("no device").
        import torch  # noqa: F401
        if os.environ.get("MM_LIB_PATH"):
            import sys
            sys.stderr.write("mimamo_net_amd: loading the library variant MM_LIB_PATH=%s\n" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, where):
    """Map mm_status codes onto the exception types the reference raises at the same spots."""
    if status == MM_OK:
        return
    if status == MM_ERR_TOO_SMALL:
        # SCFpyr_PyTorch.py:90-91
        raise RuntimeError("Cannot build the requested number of levels, image too small.")
    if status == MM_ERR_INVALID_ARG:
        raise ValueError("%s: invalid argument" % where)
    if status == MM_ERR_UNSUPPORTED:
        raise NotImplementedError("%s: %s" % (where, lib().mm_status_string(status).decode()))
    raise MimamoHipError(status, where)


def ptr(t):
    """Raw device/host address of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
