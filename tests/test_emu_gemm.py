"""bf16 GEMM kernel (csrc/s6d_gemm.hip) on the emulator: tile order, LDS ring / swizzle, counted waits, epilogue layout.
Both completion models of the LDS-DMA are run (tests/host_cc/hipemu: early = lands at the issue, late = lands when a counted
wait retires it); a schedule with a missing wait or an early restage fails one of them."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref(a, w, bias, gelu):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if gelu:
        y = torch.nn.functional.gelu(y)
    return y


def _check(ops, M, N, K, bias, gelu, max_blocks, lda_pad=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    a_full = torch.randn(M, K + lda_pad, generator=g).to(torch.bfloat16)
    a = a_full[:, :K]
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g) if bias else None
    ops.set_gemm_small_tile(False)            # these cases walk the 256 x 256 kernel's tile order and ring (round 6: the library would
    try:                                      # give such small products to the 256 x 128 kernel, covered by the run below)
        out = ops.gemm_bf16(a, w, b, gelu=gelu, max_blocks=max_blocks)
    finally:
        ops.set_gemm_small_tile(True)
    assert torch.equal(out, ops.gemm_bf16(a, w, b, gelu=gelu, max_blocks=max_blocks))      # ... and must give the same bits
    ref = _ref(a, w, b, gelu)
    err = (out.float() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-5        # one bf16 rounding of an fp32-accumulated result
    bad = (err > tol * 1.01).sum().item()
    assert bad == 0, (bad, err.max().item())


CASES = [
    # M, N, K, bias, gelu, max_blocks, lda_pad
    (256, 256, 64, False, False, 0, 0),        # one tile, one K tile (all guards of the short stream)
    (256, 256, 128, True, False, 0, 0),        # two K tiles
    (300, 256, 320, True, True, 0, 8),         # ragged M, strided A rows, ring wraps (5 K tiles), GELU epilogue
    (700, 512, 192, True, False, 8, 0),        # 6 tiles on 8 workgroups
    (1280, 768, 192, True, True, 8, 0),        # 15 tiles on 8 workgroups: persistent stream across output tiles
    (520, 384, 256, True, True, 8, 0),         # N % 256 != 0 (256 x 128 tiles only), ragged M, 9 tiles on 8 workgroups
    (1280, 768, 64, True, False, 8, 0),        # one K tile per output tile, 15 tiles on 8 workgroups: every K tile is first AND last
    (1100, 512, 128, False, True, 8, 0),       # two K tiles per output tile, 10 tiles on 8 workgroups (drain / no-wait K tiles adjacent)
]


def _run_case(i):
    from tests import hipemu  # noqa: F401
    import ctypes

    from sam6d_amd import _lib, ops
    L = ctypes.CDLL(hipemu.build())
    L.s6d_strerror.restype = ctypes.c_char_p
    L.s6d_strerror.argtypes = [ctypes.c_int]
    L.s6d_last_hip_error.restype = ctypes.c_char_p
    _lib._lib = L
    ops._stream = lambda: ctypes.c_void_p(0)
    torch.Tensor.is_cuda = property(lambda self: True)
    M, N, K, bias, gelu, mb, pad = CASES[i]
    _check(ops, M, N, K, bias, gelu, mb, pad, seed=i)


@pytest.mark.parametrize("mode", ["early", "late"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_gemm_on_the_emulator(case, mode):
    """The launcher picks the kernel by shape: N % 256 == 0 -> the eight-wave 256 x 256 kernel, N % 256 == 128 -> the kernel of two
    independent 256 x 128 workgroups per CU; CASES holds both kinds."""
    if case == 4 and not os.environ.get("S6D_EMU_SLOW") and mode == "late":
        pytest.skip("long case runs once by default (S6D_EMU_SLOW=1 for every combination)")
    # HIPEMU_GLDS is read once per process: each combination gets its own interpreter
    env = dict(os.environ, HIPEMU_GLDS=mode)
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); "
                        f"from tests import test_emu_gemm as t; t._run_case({case})"], env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _run_cblk():
    from tests import hipemu  # noqa: F401
    import ctypes

    from sam6d_amd import _lib, ops
    from tests import test_gpu_gemm as T
    L = ctypes.CDLL(hipemu.build())
    L.s6d_strerror.restype = ctypes.c_char_p
    L.s6d_strerror.argtypes = [ctypes.c_int]
    L.s6d_last_hip_error.restype = ctypes.c_char_p
    _lib._lib = L
    ops._stream = lambda: ctypes.c_void_p(0)
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    T.test_column_block_output_equals_the_plain_product(700, 768, 192, 64)
    T.test_column_block_output_equals_the_plain_product(300, 256, 64, 32)


def test_column_block_output_on_the_emulator():
    """s6d_gemm_bf16_cblk (head-major q/k/v out of the qkv GEMM) through the quad-transposed epilogue, ragged M included."""
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); "
                        f"from tests import test_emu_gemm as t; t._run_cblk()"], env=dict(os.environ, HIPEMU_GLDS="late"),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _run_res():
    from tests import hipemu  # noqa: F401
    import ctypes

    from sam6d_amd import _lib, ops
    from tests import test_gpu_gemm as T
    L = ctypes.CDLL(hipemu.build())
    L.s6d_strerror.restype = ctypes.c_char_p
    L.s6d_strerror.argtypes = [ctypes.c_int]
    L.s6d_last_hip_error.restype = ctypes.c_char_p
    _lib._lib = L
    ops._stream = lambda: ctypes.c_void_p(0)
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    T.test_residual_gemm_sums_in_the_accumulators(700, 768, 192, False)
    T.test_residual_gemm_sums_in_the_accumulators(300, 256, 64, True)
    T.test_residual_gemm_row_statistics(700, 768, 192)
    T.test_lnfold_gemm_vs_layernorm_then_linear(700, 768, 192, False, 0)
    T.test_residual_gemm_row_statistics(260, 1280, 64)      # 40 column groups: the all-loads-first finalize kernel (round 5)
    T.test_lnfold_gemm_vs_layernorm_then_linear(300, 256, 320, True, 0)
    T.test_lnfold_gemm_vs_layernorm_then_linear(520, 768, 256, False, 64)
    T.test_lnfold_gemm_with_offset_rows(300, 256, 320)
    T.test_float16_gemm_vs_float(300, 256, 320, True)
    T.test_float16_gemm_vs_float(700, 768, 192, False)
    T.test_small_tile_form_gives_the_bits_of_the_256_tile_form(300, 256, 320, True, False, 0)     # round 6: 256 x 128 tiles for under-filled launches
    T.test_small_tile_form_gives_the_bits_of_the_256_tile_form(700, 768, 192, False, False, 16)
    T.test_small_tile_form_gives_the_bits_of_the_256_tile_form(261, 512, 64, True, True, 8)
    T.test_small_tile_residual_form_gives_the_bits_of_the_256_tile_form(700, 768, 192, False, True, True)
    T.test_small_tile_residual_form_gives_the_bits_of_the_256_tile_form(300, 256, 64, True, False, True)
    T.test_small_tile_residual_form_gives_the_bits_of_the_256_tile_form(261, 1280, 128, False, True, False)


@pytest.mark.parametrize("mode", ["early", "late"])
def test_residual_epilogue_on_the_emulator(mode):
    """s6d_gemm_bf16_res (residual tile through the matrix cores + row statistics), s6d_gemm_bf16_lnfold (LayerNorm folded into
    the epilogue) and the float16 GEMM, on the host."""
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); "
                        f"from tests import test_emu_gemm as t; t._run_res()"], env=dict(os.environ, HIPEMU_GLDS=mode),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
