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


def test_layout_kernels_on_the_emulator(emu):
    TS.test_layout_kernels_bit_exact_vs_the_library_statements()


def test_crop_kernel_on_the_emulator(emu):
    TD.test_crops_bit_exact_vs_oracle_and_golden(56)
    TD.test_crops_random_boxes_bit_exact_vs_oracle()


def test_decoder_kernels_on_the_emulator(emu):
    T.test_img2tok_kernel_vs_restated_algebra()
    T.test_img2tok_raw_kernel_vs_explicit_q_projection()
    T.test_upscale_heads_kernel_vs_restated_algebra()
    T.test_tok2img_kernel_vs_oracle()
    T.test_tok2img_raw_kernel_vs_oracle_attention_with_explicit_projections()


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


def test_folded_block_loops_against_the_round2_form_and_float(emu, monkeypatch):
    """ImageEncoderViT._blocks_fused in its three forms on a 2-block encoder with a windowed and a global block (dim 256 so that
    the 256 x 256-tile kernel takes every GEMM):
      * default: residual adds through the matrix cores of proj / lin2 (s6d_gemm_bf16_res, in place, emitting row statistics) and
        both LayerNorms folded into qkv / lin1 (s6d_gemm_bf16_lnfold) -- no add_layernorm launch at all;
      * S6D_LNFOLD=0 S6D_GEMM_RES=1: residual GEMMs + one-read LayerNorm passes;
      * S6D_LNFOLD=0: round 2 (each add folded into the following LayerNorm pass).
    All three against the float32 statement of the module; the folded form must not be further from it than the round-2 form."""
    from functools import partial

    import torch

    from sam6d_amd.sam.image_encoder import ImageEncoderViT
    from sam6d_amd.utils import seeded
    m = ImageEncoderViT(depth=2, embed_dim=256, img_size=256, mlp_ratio=2, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                        num_heads=4, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=(1,), window_size=7,
                        out_chans=32).eval()
    m = seeded.load_seeded(m, 4)
    with torch.no_grad():
        for blk in m.blocks:                                          # non-trivial affine parameters: the fold has work to do
            for n in (blk.norm1, blk.norm2):
                n.weight.add_(0.2 * torch.randn(256, generator=torch.Generator().manual_seed(7)))
                n.bias.add_(0.1 * torch.randn(256, generator=torch.Generator().manual_seed(8)))
    x = (0.5 * torch.randn(1, 16, 16, 256, generator=torch.Generator().manual_seed(1)) + 0.3).to(torch.bfloat16)
    with torch.no_grad():
        ref = x.float()
        for blk in m.blocks:
            ref = blk(ref)
    m = m.bfloat16()
    keep = x.clone()
    calls = {"res": 0, "fold": 0, "ln": 0}
    real_g, real_f, real_l = emu.gemm_bf16, emu.gemm_bf16_lnfold, emu.add_layernorm
    monkeypatch.setattr(emu, "gemm_bf16", lambda *a, **k: (calls.__setitem__("res", calls["res"] + (k.get("residual") is not None)), real_g(*a, **k))[1])
    monkeypatch.setattr(emu, "gemm_bf16_lnfold", lambda *a, **k: (calls.__setitem__("fold", calls["fold"] + 1), real_f(*a, **k))[1])
    monkeypatch.setattr(emu, "add_layernorm", lambda *a, **k: (calls.__setitem__("ln", calls["ln"] + 1), real_l(*a, **k))[1])

    def rel(a):
        return ((a.float() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt().item()

    with torch.no_grad():
        fold = m._blocks_fused(x, None)
        assert calls == {"res": 4, "fold": 4, "ln": 0} and torch.equal(x, keep), calls   # the caller's tensor untouched
        monkeypatch.setenv("S6D_LNFOLD", "0")
        monkeypatch.setenv("S6D_GEMM_RES", "1")
        calls.update(res=0, fold=0, ln=0)
        res = m._blocks_fused(x, None)
        assert calls == {"res": 4, "fold": 0, "ln": 4}, calls
        monkeypatch.setenv("S6D_GEMM_RES", "0")
        calls.update(res=0, fold=0, ln=0)
        old = m._blocks_fused(x, None)
        assert calls == {"res": 0, "fold": 0, "ln": 4}, calls
    e_fold, e_res, e_old = rel(fold), rel(res), rel(old)
    print("rel rms vs float: fold %.3e res %.3e round2 %.3e" % (e_fold, e_res, e_old))
    assert e_old <= 8e-3 and e_res <= 8e-3, (e_fold, e_res, e_old)
    assert e_fold <= 1.1 * e_old + 5e-4, (e_fold, e_res, e_old)


def test_dinov2_folded_block_loop_against_float(emu, monkeypatch):
    """DinoVisionTransformer._blocks_fused (LayerScale folded into proj / fc2, residual adds and block LayerNorms folded into the GEMMs)
    on a 2-block ViT (dim 256, 4 heads of 64, 17 tokens) against the float32 statement of the module, and not further from it than
    the add + LayerNorm form (S6D_LNFOLD=0)."""
    import torch

    from sam6d_amd.ism.dinov2 import DinoVisionTransformer
    from sam6d_amd.utils import seeded
    m = DinoVisionTransformer(img_size=56, patch_size=14, embed_dim=256, depth=2, num_heads=4, mlp_ratio=4, init_values=1.0,
                              block_chunks=0).eval()
    m = seeded.load_seeded(m, 6)
    with torch.no_grad():
        for blk in m.blocks:
            for n in (blk.norm1, blk.norm2):
                n.weight.add_(0.2 * torch.randn(256, generator=torch.Generator().manual_seed(7)))
                n.bias.add_(0.1 * torch.randn(256, generator=torch.Generator().manual_seed(8)))
            blk.ls1.gamma.mul_(0.7)
            blk.ls2.gamma.mul_(1.3)
    x = (0.5 * torch.randn(2, 17, 256, generator=torch.Generator().manual_seed(1)) + 0.2).to(torch.bfloat16)
    with torch.no_grad():
        ref = x.float()
        for blk in m.blocks:
            ref = blk(ref)
        refn = m.norm(ref)
    m = m.bfloat16()
    calls = {"fold": 0, "ln": 0}
    real_f, real_l = emu.gemm_bf16_lnfold, emu.add_layernorm
    monkeypatch.setattr(emu, "gemm_bf16_lnfold", lambda *a, **k: (calls.__setitem__("fold", calls["fold"] + 1), real_f(*a, **k))[1])
    monkeypatch.setattr(emu, "add_layernorm", lambda *a, **k: (calls.__setitem__("ln", calls["ln"] + 1), real_l(*a, **k))[1])

    def rel(a, r):
        return ((a.float() - r).pow(2).mean() / r.pow(2).mean()).sqrt().item()

    with torch.no_grad():
        xf, xnf = m._blocks_fused(x)
        assert calls == {"fold": 4, "ln": 1}, calls                   # only the final norm is a LayerNorm launch
        monkeypatch.setenv("S6D_LNFOLD", "0")
        calls.update(fold=0, ln=0)
        xo, xno = m._blocks_fused(x)
        assert calls["fold"] == 0 and calls["ln"] == 5, calls
    e_f, e_o = rel(xnf, refn), rel(xno, refn)
    print("DINOv2 mini: rel rms vs float, folded %.3e, add + LayerNorm form %.3e" % (e_f, e_o))
    assert rel(xf, ref) <= 8e-3 and e_o <= 1e-2, (rel(xf, ref), e_f, e_o)
    assert e_f <= 1.1 * e_o + 5e-4, (e_f, e_o)


def test_token_side_kernels_on_the_emulator(emu):
    T.test_token_side_kernels_vs_autocast_statement(6, 7)


def test_token_side_folds_on_the_emulator(emu):
    T.test_token_side_kernels_with_the_folds_inside(2, 5)
