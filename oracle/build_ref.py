"""TEST INFRASTRUCTURE -- builds oracle/_ref/libpn2_ref_{off,fast}.so from the REFERENCE's own PointNet++ CUDA sources.

    python -m oracle.build_ref            (build container only: needs /root/reference, read-only)

The three kernel files (sampling_gpu.cu, ball_query_gpu.cu, group_points_gpu.cu under
SAM-6D/Pose_Estimation_Model/model/pointnet2/_ext_src/src/) are read where they lie and compiled for the HOST against
stand-in headers (oracle/ref_shim/: cuda.h, cuda_runtime.h, ATen/ATen.h, ATen/cuda/CUDAContext.h) on top of the emulated
runtime of tests/host_cc/hipemu (blocks sequential, threads of a block = fibers, __syncthreads = rendezvous).  The reference's
own cuda_utils.h is included from its own directory.  The only thing no host compiler accepts is the launch syntax, so the
recipe lowers `kernel<<<grid, block, shmem, stream>>>(args)` to `hipLaunchKernelGGL((kernel), grid, block, shmem, stream, args)`
-- the rewrite nvcc's front end performs -- in a scratch copy under oracle/_ref/build/ (git-ignored; `#line` directives point
back at the reference file); kernel bodies and wrappers are compiled as written.  Nothing of the reference enters the repo.

Scheduling.  The FPS kernel reads `old = dists_i[0]` after the last barrier of an iteration and thread 0 overwrites that slot
at the top of the next one with no barrier in between (sampling_gpu.cu:169-173 vs :116): harmless on a GPU, where the threads
of a block advance together, fatal under a scheduler that runs one fiber from barrier to barrier.  The kernel files are
therefore compiled with `-fsanitize-coverage=trace-pc-guard` and the guard callback (ref_shim/pn2_ref_api.cc) yields the
fiber at EVERY basic-block edge: the threads of a block advance one basic block per round-robin turn, i.e. in lockstep at the
granularity the hardware's SIMT execution guarantees at least.

Two builds, because the one thing the source does not decide is how  a*a + b*b + c*c  is contracted:
  off   -ffp-contract=off                 every product and sum rounded (nvcc --fmad=false)
  fast  -ffp-contract=fast -mfma          the host compiler's (LLVM's) contraction: fma(c,c, fma(a,a, b*b))
nvcc's default (--fmad=true) is believed to produce fma(c,c, fma(b,b, a*a)) (FMUL, FFMA, FFMA in source order); no compiler
here reproduces that from the unmodified source, so that spelling exists only in oracle/pn2_oracle.c (mode 0) and in the HIP
kernels, and stays the ONE unpinned assumption of rows a13 / a19.  tests/test_oracle_pn2_ref.py holds pn2_oracle.c to these
two libraries bit for bit in modes 2 (off) and 1 (llvm) and records where the three spellings diverge.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("S6D_REFERENCE_ROOT", "/root/reference")
EXT = os.path.join(REF, "SAM-6D", "Pose_Estimation_Model", "model", "pointnet2", "_ext_src")
OUT = os.path.join(HERE, "_ref")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FILES = ("sampling_gpu.cu", "ball_query_gpu.cu", "group_points_gpu.cu")
VARIANTS = {"off": ["-ffp-contract=off"], "fast": ["-ffp-contract=fast", "-mfma"]}


def _split_top(s):
    """Split at top-level commas (parentheses / angle brackets of template arguments respected)."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def lower_launches(text):
    """kernel<<<a, b, c, d>>>(  ->  hipLaunchKernelGGL((kernel), a, b, c, d,   (the closing parenthesis stays)."""
    pat = re.compile(r"([A-Za-z_]\w*(?:\s*<[^<>;(){}]*>)?)\s*<<<(.*?)>>>\s*\(", re.S)

    def sub(m):
        cfg = _split_top(m.group(2))
        assert len(cfg) == 4, cfg
        return f"hipLaunchKernelGGL(({m.group(1).strip()}), dim3({cfg[0]}), dim3({cfg[1]}), {cfg[2]}, {cfg[3]}, "
    out, n = pat.subn(sub, text)
    # keep the line count: a launch that spanned lines now sits on one, pad so that #line stays truthful enough for messages
    return out, n


def build(force=False):
    if not os.path.isdir(EXT):
        raise RuntimeError(f"{EXT} not found: the _ref libraries are built in the build container only")
    bdir = os.path.join(OUT, "build")
    os.makedirs(bdir, exist_ok=True)
    srcs = []
    for f in FILES:
        path = os.path.join(EXT, "src", f)
        text, n = lower_launches(open(path).read())
        assert "<<<" not in text and n > 0, f
        dst = os.path.join(bdir, f[:-3] + ".cc")
        with open(dst, "w") as g:
            g.write(f'#line 1 "{path}"\n' + text)
        srcs.append(dst)
    emu = os.path.join(REPO, "tests", "host_cc", "hipemu")
    libs = {}
    for tag, flags in VARIANTS.items():
        so = os.path.join(OUT, f"libpn2_ref_{tag}.so")
        libs[tag] = so
        deps = srcs + [os.path.join(emu, "hipemu.cc"), os.path.join(emu, "hip", "hip_runtime.h"), __file__]
        if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
            continue
        base = [CXX, "-O2", "-std=c++17", "-fPIC", "-w"] + flags + ["-I", os.path.join(HERE, "ref_shim"), "-I", emu, "-I",
                os.path.join(EXT, "include")]
        objs = []
        for src, extra in [(x, ["-fsanitize-coverage=trace-pc-guard"]) for x in srcs] + \
                [(os.path.join(emu, "hipemu.cc"), []), (os.path.join(HERE, "ref_shim", "pn2_ref_api.cc"), [])]:
            obj = os.path.join(bdir, f"{tag}_{os.path.basename(src)}.o")
            r = subprocess.run(base + extra + ["-c", src, "-o", obj], capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"pn2 _ref build failed ({src}):\n" + r.stderr[-4000:])
            objs.append(obj)
        r = subprocess.run([CXX, "-shared", "-o", so] + objs, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("pn2 _ref link failed:\n" + r.stderr[-4000:])
    return libs


if __name__ == "__main__":
    for k, v in build(force="-f" in sys.argv).items():
        print(k, v)
