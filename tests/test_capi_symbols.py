"""The C-ABI library loads on a CPU-only host and exports every symbol include/sam6d_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes

from sam6d_amd import _lib


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.s6d_version() >= 100
    assert L.s6d_strerror(-1).decode().startswith("invalid argument")


def test_argument_validation_without_a_gpu():
    """Entry points validate sizes/pointers before touching the device: callable on a CPU-only host."""
    L = _lib.lib()
    L.s6d_fps_f32.restype = ctypes.c_int
    assert L.s6d_fps_f32(None, 1, 0, 1, None, None, None) == -1          # N <= 0
    assert L.s6d_fps_f32(None, 0, 10, 4, None, None, None) == 0           # B == 0: nothing to do
    assert L.s6d_fps_f32(None, 1, 10, 4, None, None, None) == -1          # null pointers
    L.s6d_rpe_attention_f32.restype = ctypes.c_int
    assert L.s6d_rpe_attention_f32(None, None, None, None, None, None, 1, 197, 128, 4, ctypes.c_float(1.0), None, None) == -3


def test_round3_entry_points_validate_shapes_without_a_gpu():
    """The folded-GEMM, statistics and linear-attention entries reject what they do not implement before any launch (fake non-null
    pointers: nothing is dereferenced on these paths)."""
    L = _lib.lib()
    p = ctypes.c_void_p(4096)
    lng, flt = ctypes.c_long, ctypes.c_float
    for fn in ("s6d_gemm_bf16_lnfold", "s6d_gemm_bf16_res", "s6d_ln_stats_finalize", "s6d_row_stats_bf16", "s6d_linear_attention_f32",
               "s6d_rpe_attention_strided_f32", "s6d_mha_strided_f32"):
        getattr(L, fn).restype = ctypes.c_int
    # N % 256 != 0: the 256 x 256-tile kernel only
    assert L.s6d_gemm_bf16_lnfold(p, lng(64), p, p, lng(64), p, p, p, lng(128), 256, 128, 64, 0, 0, 0, None) == -3
    assert L.s6d_gemm_bf16_lnfold(p, lng(64), p, p, lng(64), p, p, p, lng(256), 256, 256, 64, 2, 0, 0, None) == -1       # gelu flag
    assert L.s6d_gemm_bf16_res(p, lng(64), p, lng(64), p, p, lng(128), None, p, lng(128), 256, 128, 64, 0, None) == -3
    assert L.s6d_gemm_bf16_res(p, lng(64), p, lng(64), p, p, lng(100), None, p, lng(256), 256, 256, 64, 0, None) == -1     # ldr < N
    assert L.s6d_ln_stats_finalize(p, 0, 32, lng(16), flt(1e-6), p, None) == -1
    assert L.s6d_ln_stats_finalize(p, 40, 32, lng(0), flt(1e-6), p, None) == 0                                               # no rows
    assert L.s6d_row_stats_bf16(p, lng(1284), lng(4), 1284, flt(1e-6), p, None) == -1                                        # C % 8
    assert L.s6d_row_stats_bf16(p, lng(2048), lng(4), 2048, flt(1e-6), p, None) == -3                                        # C > 1536
    assert L.s6d_linear_attention_f32(p, p, 3, p, lng(128), p, lng(128), 1, 8, 8, 128, p, p, None) == -3                     # C != 256
    assert L.s6d_linear_attention_f32(p, p, 3, p, lng(100), p, lng(256), 1, 8, 8, 256, p, p, None) == -1                     # ldk < C
    assert L.s6d_rpe_attention_strided_f32(p, lng(258), p, lng(256), p, lng(256), p, p, p, 1, 8, 256, 4, flt(1.0), p, None) == -1   # ld % 4
    assert L.s6d_mha_strided_f32(p, lng(768), p, lng(512), p, lng(512), 0, 8, 8, 256, 4, flt(1.0), p, None) == 0             # B == 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from sam6d_amd import ops
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.ball_query(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), 0.1, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.rpe_attention(*([torch.zeros(1, 4, 256)] * 3), torch.zeros(1, 4, 4, 256), torch.zeros(1, 4, 4),
                          torch.zeros(1, 4, 4, 256), 0.125)


def test_every_entry_point_refuses_null_operands():
    """Generated from the header: each `int s6d_*(...)` prototype is called with NULL for every pointer and small positive
    sizes.  Every one must come back with S6D_EINVAL / S6D_EUNSUPPORTED before anything is dereferenced or launched
    (this host has no device, so a launch attempt would surface as S6D_ELAUNCH = -2)."""
    import os
    import re

    L = _lib.lib()
    hdr = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "sam6d_hip.h")
    src = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    protos = re.findall(r"\bint\s+(s6d_\w+)\s*\(([^)]*)\)\s*;", src)
    assert len(protos) >= 28
    kinds = {"float": lambda: ctypes.c_float(1.0), "double": lambda: ctypes.c_double(1.0), "long": lambda: ctypes.c_long(16),
             "int": lambda: ctypes.c_int(16)}
    rcs = {}
    for name, args in protos:
        if args.strip() == "void" or "*" not in args:          # no operand to refuse (s6d_version, s6d_set_persistent_grid_limit)
            continue
        vals = []
        for a in (x.strip() for x in args.split(",")):
            vals.append(ctypes.c_void_p(0) if "*" in a else kinds[a.split()[0]]())
        fn = getattr(L, name)
        fn.restype = ctypes.c_int
        rcs[name] = fn(*vals)
    assert all(rc in (-1, -3) for rc in rcs.values()), rcs


def test_product_library_carries_no_emulator_code():
    """The kernel sources have `#ifdef HIPEMU` blocks for the host emulator (tests/host_cc/hipemu).  HIPEMU is defined by that
    emulator's stand-in <hip/hip_runtime.h> only, so the product build cannot see those branches: the hipcc preprocessor does not
    define it for gfx950, and libsam6d_hip.so contains neither the emulator's namespace nor its abort messages."""
    import os
    import subprocess

    from sam6d_amd import _lib
    if os.path.exists(_lib.HIPCC):
        out = subprocess.run([_lib.HIPCC, f"--offload-arch={_lib.ARCH}", "-dM", "-E", "-x", "hip", "/dev/null"], capture_output=True,
                             text=True).stdout
        assert "__gfx950__" in out and "HIPEMU" not in out
    blob = open(_lib.SO_PATH, "rb").read()
    assert b"hipemu" not in blob and b"HIPEMU" not in blob


def test_gemm4_compiler_code_leaves_the_accumulator_file_alone(tmp_path):
    """csrc/s6d_gemm4.hip keeps its 256 accumulators in a[0:255] across inline-asm blocks the compiler knows nothing about: that is
    sound only while the compiler-generated code around the blocks never touches the accumulator file and never spills.  Checked on
    the generated assembly of every instantiation: outside ;;#ASMSTART / ;;#ASMEND no instruction names an a-register, there is no
    scratch access, and the kernel descriptors reserve the whole accumulator file behind the architected registers."""
    import os
    import re
    import subprocess
    src = os.path.join(os.path.dirname(_lib.__file__), "csrc", "s6d_gemm4.hip")
    out = tmp_path / "g4.s"
    flags = [f for f in _lib.FLAGS if f not in ("-shared", "-fPIC")] + _lib.file_flags(src)
    subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", str(out), src], cwd=os.path.dirname(src))
    text = out.read_text()
    kernels = re.findall(r"^(_ZN3s6d\w*gemm4_\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.M | re.S)
    assert len(kernels) >= 10
    for name, body in kernels:
        inside = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                inside = True
            elif "#ASMEND" in line:
                inside = False
            elif not inside and not line.strip().startswith(";"):
                assert not re.search(r"\ba\[?\d+|accvgpr", line), f"{name}: compiler code touches the accumulator file: {line.strip()}"
                assert "scratch_" not in line, f"{name}: spill: {line.strip()}"
    for m in re.finditer(r"\.amdhsa_kernel (\S*gemm4\S*)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        nxt = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1))
        off = int(re.search(r"\.amdhsa_accum_offset (\d+)", m.group(2)).group(1))
        assert nxt - off == 256 and nxt <= 512, (m.group(1), nxt, off)
