"""GPU parity of the drop-in SAM ImageEncoderViT vs reference goldens / the CPU oracle."""
from functools import partial

import numpy as np
import pytest
import torch

from oracle import sam as osam
from sam6d_amd.utils import seeded, synth
from tests import util

pytestmark = pytest.mark.gpu


def _mini():
    from sam6d_amd.sam.image_encoder import ImageEncoderViT
    c = osam.MINI
    return ImageEncoderViT(depth=c["depth"], embed_dim=c["dim"], img_size=c["img_size"], mlp_ratio=4,
                           norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=c["heads"], patch_size=16,
                           qkv_bias=True, use_rel_pos=True, global_attn_indexes=c["global_idx"],
                           window_size=c["window"], out_chans=c["out_chans"]).eval()


def test_mini_encoder_fp32_vs_reference_golden(monkeypatch):
    monkeypatch.setenv("S6D_SAM_DTYPE", "fp32")
    g = util.golden("sam_enc.npz")
    m = seeded.load_seeded(_mini(), 3).cuda()
    with torch.no_grad():
        y = m(synth.sam_input(1, 5, osam.MINI["img_size"]).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g["mini_out"], rtol=1e-3, atol=2e-4)


def test_mini_encoder_bf16_vs_reference_golden(monkeypatch):
    """bf16 compute (the bench dtype): tolerance = bf16 round-off through 4 blocks."""
    monkeypatch.setenv("S6D_SAM_DTYPE", "bf16")
    g = util.golden("sam_enc.npz")
    m = seeded.load_seeded(_mini(), 3).cuda()
    with torch.no_grad():
        y = m(synth.sam_input(1, 5, osam.MINI["img_size"]).cuda()).float().cpu().numpy()
    err = np.abs(y - g["mini_out"])
    assert err.mean() < 2e-2 and np.corrcoef(y.ravel(), g["mini_out"].ravel())[0, 1] > 0.999, (err.mean(), err.max())


def test_vit_h_two_blocks_and_full_vs_reference_golden(monkeypatch):
    from sam6d_amd.sam.image_encoder import build_vit_h
    g = util.golden("sam_enc.npz")
    m = seeded.load_seeded(build_vit_h().eval(), 3).cuda()
    x = synth.sam_input(1, 5, 1024).cuda()
    monkeypatch.setenv("S6D_SAM_DTYPE", "fp32")
    with torch.no_grad():
        t = m.forward_tokens(x, upto=2)
    util.assert_digest_close(t, g["h_blk2_sum"], g["h_blk2_smp"], 1009, 1e-3, 1e-4, "vit-h 2 blocks fp32")
    monkeypatch.setenv("S6D_SAM_DTYPE", "bf16")
    with torch.no_grad():
        y = m(x).float().cpu()
    smp = y.reshape(-1)[::251].numpy()
    assert np.corrcoef(smp, g["h_smp"])[0, 1] > 0.995, np.corrcoef(smp, g["h_smp"])[0, 1]
    assert np.abs(smp - g["h_smp"]).mean() < 5e-2


def test_preprocess_matches_oracle():
    from sam6d_amd.sam.image_encoder import preprocess
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 768, 1024, generator=g) * 255
    ref = torch.stack([osam.preprocess(x[i]) for i in range(2)])
    out32 = preprocess(x.cuda(), out_dtype=torch.float32).cpu()
    assert out32.shape == (2, 3, 1024, 1024) and (out32 - ref).abs().max() < 1e-5
    out16 = preprocess(x.cuda(), out_dtype=torch.bfloat16).float().cpu()
    assert (out16 - ref).abs().max() < 2e-2 and (out16[:, :, 768:] == 0).all()
