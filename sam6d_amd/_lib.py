"""Build + load libsam6d_hip.so (the C-ABI library declared in include/sam6d_hip.h).

The library is built in-tree with hipcc for gfx950 (cross-compiles without a GPU) and
loaded with ctypes.  There is NO CPU fallback: if the library is missing or fails to
load, importing any op raises.
"""
import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libsam6d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950 has a unified register file); removes the
# v_accvgpr_read/write traffic around every softmax rescale (+6 % on the attention kernels, measured).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]

_lib = None


def sources():
    return sorted(glob.glob(os.path.join(_CSRC, "*.hip")))


def file_flags(src):
    """Extra compiler flags of ONE source: a first line `// hipcc-flags: ...` (csrc/s6d_gemm4.hip switches the SLP vectoriser off)."""
    with open(src) as f:
        first = f.readline()
    return first.split(":", 1)[1].split() if first.startswith("// hipcc-flags:") else []


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = sources() + glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libsam6d_hip.so.  One object per source under csrc/build/ (git-ignored),
    compiled in parallel and only when the source or a header is newer: a change in one kernel file rebuilds in seconds."""
    if not force and not _stale():
        return SO_PATH
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build libsam6d_hip.so")
    objdir = os.path.join(_CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time: `bench.py --gpus N` starts N ranks that may all find the library stale; they share the object directory,
    # so the build runs under an exclusive file lock and whoever waited re-checks staleness instead of compiling again
    import fcntl
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return SO_PATH
            return _build_locked(force, verbose, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, objdir):
    from concurrent.futures import ThreadPoolExecutor
    extra = os.environ.get("S6D_EXTRA_HIPCC_FLAGS", "").split()
    cflags = [f for f in FLAGS if f != "-shared"] + extra
    hdrs = glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    # s6d_attn_f16.hip re-compiles s6d_attn.hip under another element type: it depends on that source too
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    tag = os.path.join(objdir, ".flags")
    flags_now = " ".join(cflags)
    if not os.path.exists(tag) or open(tag).read() != flags_now:
        force = True

    def one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        dep_t = max(hdr_t, os.path.getmtime(src), max(os.path.getmtime(x) for x in sources()) if src.endswith("_f16.hip") else 0)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < dep_t:
            cmd = [HIPCC] + cflags + file_flags(src) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=_CSRC)
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, sources()))
    open(tag, "w").write(flags_now)
    tmp = SO_PATH + f".{os.getpid()}.tmp"                       # link beside the target, then rename: a reader never sees a half-written .so
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    os.replace(tmp, SO_PATH)
    return SO_PATH


def lib():
    """The loaded library.  torch is imported first so that the HIP runtime torch ships
    (same soname libamdhip64.so.7) is the one this library binds to: streams and device
    pointers are then shared with torch."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (must precede the CDLL)

        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP extension is mandatory; there is no fallback path)")
        L = ctypes.CDLL(SO_PATH)
        L.s6d_strerror.restype = ctypes.c_char_p
        L.s6d_strerror.argtypes = [ctypes.c_int]
        L.s6d_last_hip_error.restype = ctypes.c_char_p
        L.s6d_version.restype = ctypes.c_int
        want = abi_version()
        if L.s6d_version() != want:
            raise RuntimeError(f"{SO_PATH} is ABI version {L.s6d_version()}, include/sam6d_hip.h is {want}: rebuild "
                               "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = L
    return _lib


class S6DError(RuntimeError):
    pass


def check(code, what):
    if code != 0:
        L = lib()
        raise S6DError(f"{what}: {L.s6d_strerror(code).decode()} [{L.s6d_last_hip_error().decode()}]")


def abi_version():
    """S6D_ABI_VERSION of include/sam6d_hip.h."""
    import re

    return int(re.search(r"#define\s+S6D_ABI_VERSION\s+(\d+)", open(os.path.join(_HERE, "..", "include", "sam6d_hip.h")).read()).group(1))


def declared_symbols():
    """Entry points declared in include/sam6d_hip.h (used by the export test)."""
    import re

    hdr = open(os.path.join(_HERE, "..", "include", "sam6d_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(s6d_[a-z0-9_]+)\s*\(", hdr)))
