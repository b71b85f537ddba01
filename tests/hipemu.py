"""Host build of the kernel sources against the emulated HIP runtime (tests/host_cc/hipemu): libsam6d_emu.so exports the same
C ABI as libsam6d_hip.so but takes HOST pointers and runs every launch on the CPU (lanes = fibers, wave collectives and MFMA
emulated).  TEST INFRASTRUCTURE -- slow, for small shapes; lets kernel logic be checked against the oracle without a GPU."""
import ctypes
import glob
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "sam6d_amd", "csrc")
EMU = os.path.join(HERE, "host_cc", "hipemu")
OUT = os.path.join(HERE, "host_cc", "_build")
# HIPEMU_EXTRA: extra compiler flags (e.g. "-DS6D_GEMM_ROLL=1" for a kernel variant); each setting gets its own library file
EXTRA = os.environ.get("HIPEMU_EXTRA", "").split()
SO = os.path.join(OUT, "libsam6d_emu" + ("_" + re.sub(r"[^A-Za-z0-9]+", "_", "".join(EXTRA)) if EXTRA else "") + ".so")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
DYN_NAMES = ("smem", "ps_smem", "sd_smem", "t2i_smem", "t2r_smem", "gemm_smem", "fm_smem", "cs_smem", "pc_smem", "tk_smem")

_lib = None


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(EMU, "**", "*"), recursive=True) + [__file__]
    return any(os.path.isfile(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, files=None):
    if not force and not _stale() and files is None:
        return SO
    os.makedirs(OUT, exist_ok=True)
    srcs = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        if files is not None and os.path.basename(f) not in files:
            continue
        text = open(f).read()
        if os.path.basename(f) == "s6d_attn_f16.hip":
            # the IEEE-half build of the attention kernels = s6d_attn.hip under S6D_ATTN_F16: include the TEXT (the transformations
            # below must reach it) instead of the #include
            text = "#define S6D_ATTN_F16 1\n" + open(os.path.join(CSRC, "s6d_attn.hip")).read()
        # dynamic LDS: `extern __shared__ ... char name[];` refers to a global array defined below
        text = text.replace("extern __shared__", "extern")
        # places that rely on a wave executing in lockstep are marked with a comment in the product source; lanes are
        # independent fibers here, so the marker becomes a real rendezvous of the wave
        text = re.sub(r"^[ \t]*// hipemu: wave rendezvous[^\n]*$", "hipemu::wave_barrier();", text, flags=re.M)
        dst = os.path.join(OUT, os.path.basename(f)[:-4] + ".cc")
        with open(dst, "w") as g:
            g.write(f'#line 1 "{f}"\n' + text)
        srcs.append(dst)
    dyn = os.path.join(OUT, "_dyn_shared.cc")
    with open(dyn, "w") as g:
        g.write("namespace s6d {\n" + "".join(f"alignas(64) char {n}[160 * 1024];\n" for n in DYN_NAMES) + "}\n"
                "namespace s6d_h {\nalignas(64) char smem[160 * 1024];\n}\n")
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-ffp-contract=off"] + EXTRA + ["-I", EMU, "-I", CSRC, "-I",
           os.path.join(REPO, "include"), "-o", SO, os.path.join(EMU, "hipemu.cc"), dyn] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("hipemu build failed:\n" + r.stderr[-6000:])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def ptr(a):
    """numpy array (C-contiguous) -> void*"""
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)
