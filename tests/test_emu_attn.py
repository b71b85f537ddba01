"""The MFMA attention kernels and the fused residual + LayerNorm, executed on the HOST through the emulated HIP runtime
(tests/hipemu.py) at small shapes: the bodies of the device parity tests of tests/test_gpu_attn.py, unchanged."""
import pytest

from tests import test_gpu_attn as T


@pytest.mark.parametrize("B,H,nh,hd,ws", [(1, 16, 1, 64, 7), (1, 16, 1, 64, 0), (1, 14, 1, 80, 14)])
def test_fused_attention_on_the_emulator(emu, B, H, nh, hd, ws):
    T.test_fused_attention_vs_library_statement(B, H, nh, hd, ws)


@pytest.mark.parametrize("rows,C", [(37, 160), (5, 768)])
def test_add_layernorm_on_the_emulator(emu, rows, C):
    T.test_add_layernorm_bf16(rows, C)
