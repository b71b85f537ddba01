"""fp8 GEMM and LayerNorm -> fp8 kernels on the emulator (scaled 32x32x64 fp8 matrix instruction and v_cvt_pk_fp8_f32 emulated in
tests/host_cc/hipemu): the bodies of tests/test_gpu_fp8.py at small shapes, both LDS-DMA completion models."""
import os
import subprocess
import sys

import pytest

from tests import test_gpu_fp8 as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_quantiser():
    T.test_row_quantiser_roundtrip_properties()


def test_layernorm_fp8_on_the_emulator(emu):
    for rows, C in ((100, 1280), (37, 160), (5, 768)):
        T.test_layernorm_fp8_vs_library_statement(rows, C)


def _run():
    import ctypes

    import torch

    from sam6d_amd import _lib, ops
    from tests import hipemu
    L = ctypes.CDLL(hipemu.build())
    L.s6d_strerror.restype = ctypes.c_char_p
    L.s6d_strerror.argtypes = [ctypes.c_int]
    L.s6d_last_hip_error.restype = ctypes.c_char_p
    _lib._lib = L
    ops._stream = lambda: ctypes.c_void_p(0)
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    for case in ((256, 256, 128, False, False, 0), (300, 256, 384, True, False, 0), (700, 512, 256, True, True, 8),
                 (1280, 768, 128, True, False, 8)):
        T.test_gemm_fp8_vs_float_on_the_quantised_operands(*case)
    # MX forms (round 4): block scales on the activations (several K tiles, two output tiles per workgroup: the scale cursor crosses a
    # tile boundary), and the GELU epilogue that writes them
    for case in ((256, 256, 128, 0), (512, 512, 384, 8), (1024, 256, 1280, 8), (1280, 768, 256, 8)):    # the last: 15 tiles on 8 workgroups
        T.test_gemm_fp8_mx_activations_vs_float(*case)
    for case in ((256, 256, 128, 0), (300, 512, 256, 8)):
        T.test_gemm_fp8_gelu_mx_output_vs_float(*case)


@pytest.mark.parametrize("mode", ["early", "late"])
def test_gemm_fp8_on_the_emulator(mode):
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from tests import test_emu_fp8 as t; t._run()"],
                       env=dict(os.environ, HIPEMU_GLDS=mode), capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_fp8_block_loop_on_the_emulator(emu, monkeypatch):
    """ImageEncoderViT._blocks_fp8 (qkv and lin1 on the fp8 kernels, LayerNorm emitting e4m3 rows) on a 2-block, 256-wide encoder
    against the bf16 block loop: within the probe's per-block budget (tests/test_gpu_fp8.py E_BLOCK_FP8) in quadrature."""
    from functools import partial

    import torch

    from sam6d_amd.sam.image_encoder import ImageEncoderViT
    from sam6d_amd.utils import seeded
    m = ImageEncoderViT(depth=2, embed_dim=256, img_size=256, mlp_ratio=2, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                        num_heads=4, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=(1,), window_size=7,
                        out_chans=32).eval()
    m = seeded.load_seeded(m, 4).bfloat16()
    x = (0.5 * torch.randn(1, 16, 16, 256, generator=torch.Generator().manual_seed(1))).to(torch.bfloat16)
    calls = []
    real = emu.gemm_fp8
    monkeypatch.setattr(emu, "gemm_fp8", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        ref = m._blocks_fused(x, None).float()
        monkeypatch.setenv("S6D_SAM_GEMM", "fp8")
        out = m._blocks_fused(x, None).float()
    assert len(calls) == 4
    rel = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rel <= T.E_BLOCK_FP8 * 3 ** 0.5, rel
    # fp8mx (round 4): lin1 -> MX e4m3 -> lin2 on the fp8 kernels, lin2's delta added by the next quantising LayerNorm
    mx = []
    real_mx = emu.gemm_fp8_mxa
    monkeypatch.setattr(emu, "gemm_fp8_mxa", lambda *a, **k: (mx.append(1), real_mx(*a, **k))[1])
    with torch.no_grad():
        monkeypatch.setenv("S6D_SAM_GEMM", "fp8mx")
        out_mx = m._blocks_fused(x, None).float()
    assert len(mx) == 2 and len(calls) == 6                          # per block: qkv on gemm_fp8, lin1 on gemm_fp8_gelu_mx, lin2 on gemm_fp8_mxa
    rel = ((out_mx - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert rel <= T.E_BLOCK_FP8 * 3 ** 0.5, rel
