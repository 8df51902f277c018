"""Build libmimamo_hip.so IN-TREE with hipcc for gfx950 (no JIT cache: the .so must travel with the
repo snapshot to the GPU box).  `python -m mimamo_net_amd.build` or __graft_entry__.build()."""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmimamo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(os.path.dirname(HERE), "include")] + os.environ.get("MM_EXTRA_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _file_flags(src):
    """Per-file compiler flags: a source may carry a line `// mm-hipcc-flags: <flags>` near its top (e.g. a backend option one of its
    kernels needs)."""
    out = []
    with open(src) as f:
        for i, line in enumerate(f):
            if i > 80:
                break
            if "mm-hipcc-flags:" in line:
                out += line.split("mm-hipcc-flags:", 1)[1].split()
    return out


def _compile(src, headers):
    obj = os.path.splitext(src)[0] + ".o"
    if _stale(obj, [src] + headers):
        cmd = [HIPCC] + FLAGS + _file_flags(src) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    return obj


def build_library(force=False, verbose=False):
    srcs = _sources()
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    if force:
        for s in srcs:
            o = os.path.splitext(s)[0] + ".o"
            if os.path.exists(o):
                os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, headers), srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    if verbose:
        print("built", LIB, os.path.getsize(LIB), "bytes from", len(objs), "objects")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
