"""SAM-side kernels executed on the HOST through the emulated HIP runtime: bodies of the device tests of
tests/test_gpu_sam_decoder.py / test_gpu_sam.py / test_gpu_dinov2.py that fit the emulator's speed (mask post-processing,
NMS, pre-processing, crop kernel bit-exact; the three fused decoder kernels against their library-op statements)."""
import pytest

from tests import test_gpu_dinov2 as TD
from tests import test_gpu_sam as TS
from tests import test_gpu_sam_decoder as T


def test_mask_post_kernel_on_the_emulator(emu):
    T.test_mask_post_kernel_bit_exact_vs_oracle_and_golden()


def test_nms_kernel_on_the_emulator(emu):
    T.test_nms_kernel_vs_torchvision_algorithm()


def test_sam_preprocess_on_the_emulator(emu):
    TS.test_preprocess_matches_oracle()


def test_crop_kernel_on_the_emulator(emu):
    TD.test_crops_bit_exact_vs_oracle_and_golden(56)
    TD.test_crops_random_boxes_bit_exact_vs_oracle()


def test_decoder_kernels_on_the_emulator(emu):
    T.test_img2tok_kernel_vs_restated_algebra()
    T.test_upscale_heads_kernel_vs_restated_algebra()
    T.test_tok2img_kernel_vs_oracle()


def test_mini_encoder_bf16_fused_path_on_the_emulator(emu, monkeypatch):
    """The whole mini SAM encoder through the fused bf16 path (window16 / global MFMA attention and fused residual + LayerNorm
    kernels emulated, GEMMs by the host library) against the reference golden, with the device test's tolerance.  The device
    path casts weights per op under CUDA autocast; without CUDA the weights are cast once, which is the same arithmetic."""
    import numpy as np
    import torch

    from oracle import sam as osam
    from sam6d_amd.utils import seeded, synth
    from tests import util
    monkeypatch.setenv("S6D_SAM_DTYPE", "bf16")
    g = util.golden("sam_enc.npz")
    m = seeded.load_seeded(TS._mini(), 3).bfloat16()
    called = []
    real = emu.window_attention
    monkeypatch.setattr(emu, "window_attention", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y = m(synth.sam_input(1, 5, osam.MINI["img_size"])).float().numpy()
    assert len(called) == osam.MINI["depth"]                      # every block went through the fused attention kernel
    err = np.abs(y - g["mini_out"])
    assert err.mean() < 2e-2 and np.corrcoef(y.ravel(), g["mini_out"].ravel())[0, 1] > 0.999, (err.mean(), err.max())
