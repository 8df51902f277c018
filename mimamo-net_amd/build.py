"""Build libmimamo_hip.so IN-TREE with hipcc for gfx950 (no JIT cache: the .so must travel with the
repo snapshot to the GPU box).  `python -m mimamo_net_amd.build` or __graft_entry__.build().

Objects live in `csrc/obj-<hash of the flag set>/`, and the library records the flag set it was linked from in
`libmimamo_hip.so.flags`: a build with other flags (tools/ scripts add `-DMM_MEASURE` through MM_EXTRA_HIPCC_FLAGS) neither reuses
the shipped objects nor is silently left behind as the shipped library -- `build_library()` relinks when the stamp differs."""
import concurrent.futures
import glob
import hashlib
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmimamo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
BASE_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
              "-I" + os.path.join(os.path.dirname(HERE), "include")]


def _flags():
    return BASE_FLAGS + os.environ.get("MM_EXTRA_HIPCC_FLAGS", "").split()


def _flag_key(flags):
    return hashlib.sha256(" ".join(flags).encode()).hexdigest()[:10]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_PROBED = {}


def _flag_supported(flags):
    """Optional backend flags (`-mllvm ...`) are internal LLVM options that a later hipcc may rename: compile an empty translation
    unit with them once; a flag set the compiler refuses is dropped with a warning (the flag is a performance matter -- register
    promotion of one kernel instantiation -- not a correctness one)."""
    key = tuple(flags)
    if key not in _PROBED:
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void mm_probe() {}\n")
            r = subprocess.run([HIPCC, "--offload-arch=" + ARCH, "-c", src, "-o", os.path.join(d, "probe.o")] + list(flags),
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        _PROBED[key] = r.returncode == 0
        if not _PROBED[key]:
            sys.stderr.write("mimamo_net_amd.build: hipcc does not accept %s -- dropped (%s)\n" % (" ".join(flags), r.stdout.strip()[:200]))
    return _PROBED[key]


def _file_flags(src):
    """Per-file compiler flags: a source may carry a line `// mm-hipcc-flags: <flags>` near its top (e.g. a backend option one of its
    kernels needs).  Kept only when the compiler accepts them (_flag_supported)."""
    out = []
    with open(src) as f:
        for i, line in enumerate(f):
            if i > 80:
                break
            if "mm-hipcc-flags:" in line:
                out += line.split("mm-hipcc-flags:", 1)[1].split()
    return out if (not out or _flag_supported(out)) else []


def _compile(src, headers, flags, objdir):
    obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
    if _stale(obj, [src] + headers):
        cmd = [HIPCC] + flags + _file_flags(src) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    return obj


def library_flags(lib=None):
    """The flag set the library at `lib` (default: the in-tree one) was built with, or None when it carries no stamp."""
    try:
        with open((lib or LIB) + ".flags") as f:
            return f.read().strip()
    except OSError:
        return None


def build_library(force=False, verbose=False, out=None):
    """Compile what is stale and link.  out: another library path (tools/ A/B variants); the default is the shipped library."""
    lib = out or LIB
    flags = _flags()
    stamp = " ".join(flags)
    srcs = _sources()
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    objdir = os.path.join(CSRC, "obj-" + _flag_key(flags))
    os.makedirs(objdir, exist_ok=True)
    if force:
        for o in glob.glob(os.path.join(objdir, "*.o")):
            os.remove(o)
    keep = {os.path.splitext(os.path.basename(s))[0] + ".o" for s in srcs}
    for o in glob.glob(os.path.join(objdir, "*.o")):          # objects of sources that no longer exist
        if os.path.basename(o) not in keep:
            os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, headers, flags, objdir), srcs))
    if force or _stale(lib, objs) or library_flags(lib) != stamp:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
        with open(lib + ".flags", "w") as f:
            f.write(stamp + "\n")
    if verbose:
        print("built", lib, os.path.getsize(lib), "bytes from", len(objs), "objects;", "flags:", stamp)
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
