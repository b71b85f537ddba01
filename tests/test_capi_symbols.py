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
        if args.strip() == "void":
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
