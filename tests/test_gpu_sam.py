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
    # the full 32-block fp32 path on the device against the reference's fp32 output (sample of the golden)
    with torch.no_grad():
        y32 = m(x).float().cpu()
    smp = y32.reshape(-1)[::251].numpy()
    np.testing.assert_allclose(smp, g["h_smp"], rtol=1e-3, atol=1e-3 * np.abs(g["h_smp"]).max())


E_BLOCK = 4e-3       # per-block rms rounding of the bf16 path relative to the block's output rms (test_vit_h_block_bf16_* bound)


def test_vit_h_bf16_error_growth_model(monkeypatch):
    """The whole ViT-H in the benched dtype, 1024^2 input, held to an ERROR MODEL instead of a correlation.
    One block of the bf16 path adds an rms error of at most E_BLOCK of its output rms (element-level test above, measured
    2.7e-3); the blocks' roundings are independent, the residual stream carries them forward, so after k blocks the token map
    must sit within E_BLOCK * sqrt(k + 1) * GAIN of the fp32 path's (rms over the map, relative to the fp32 rms; "+1" = the
    bf16 patch embedding; GAIN = 1: measured 4.3e-3 at k = 1 and 1.3e-2 at k = 32, profiles/r03_parity_margins_*.jsonl).  Checked at
    k = 1, 2, 4, 8, 16, 32 against the SAME model in fp32 on the device (pinned to the reference golden by the test above),
    then once more on the neck's output, whose two LayerNorm2d renormalise the map (same bound)."""
    from sam6d_amd.sam.image_encoder import build_vit_h
    m = seeded.load_seeded(build_vit_h().eval(), 3).cuda()
    x = synth.sam_input(1, 5, 1024).cuda()
    GAIN = 1.0
    rel = {}
    for k in (1, 2, 4, 8, 16, 32):
        with torch.no_grad():
            t32 = m.forward_tokens(x, upto=k).float()
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                t16 = m.forward_tokens(x.to(torch.bfloat16), upto=k).float()
        rel[k] = ((t16 - t32).pow(2).mean().sqrt() / t32.pow(2).mean().sqrt()).item()
    monkeypatch.setenv("S6D_SAM_DTYPE", "fp32")
    with torch.no_grad():
        y32 = m(x).float()
    monkeypatch.setenv("S6D_SAM_DTYPE", "bf16")
    with torch.no_grad():
        y16 = m(x).float()
    rel["neck"] = ((y16 - y32).pow(2).mean().sqrt() / y32.pow(2).mean().sqrt()).item()
    util.record_margin("vit_h_bf16_error_growth", **{f"rel_{k}": v for k, v in rel.items()})
    for k in (1, 2, 4, 8, 16, 32):
        assert rel[k] <= E_BLOCK * (k + 1) ** 0.5 * GAIN, rel
    assert rel["neck"] <= E_BLOCK * 33 ** 0.5 * GAIN, rel


def test_vit_h_folded_block_loop_is_what_runs(monkeypatch):
    """At the released ViT-H dimensions the bf16 block loop is the folded one (no add_layernorm launch between the blocks: two
    residual GEMMs and two LayerNorm-folded GEMMs per block), and its token map agrees with the add + LayerNorm form
    (S6D_LNFOLD=0) to the two paths' independent bf16 roundings."""
    from sam6d_amd import ops
    from sam6d_amd.sam.image_encoder import build_vit_h
    m = seeded.load_seeded(build_vit_h().eval(), 3).cuda()
    x = synth.sam_input(1, 5, 1024).cuda().to(torch.bfloat16)
    calls = {"fold": 0, "res": 0, "ln": 0}
    real_f, real_g, real_l = ops.gemm_bf16_lnfold, ops.gemm_bf16, ops.add_layernorm
    monkeypatch.setattr(ops, "gemm_bf16_lnfold", lambda *a, **k: (calls.__setitem__("fold", calls["fold"] + 1), real_f(*a, **k))[1])
    monkeypatch.setattr(ops, "gemm_bf16", lambda *a, **k: (calls.__setitem__("res", calls["res"] + (k.get("residual") is not None)), real_g(*a, **k))[1])
    monkeypatch.setattr(ops, "add_layernorm", lambda *a, **k: (calls.__setitem__("ln", calls["ln"] + 1), real_l(*a, **k))[1])
    k = 4
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        folded = m.forward_tokens(x, upto=k).float()
        assert calls == {"fold": 2 * k, "res": 2 * k, "ln": 0}, calls
        monkeypatch.setenv("S6D_LNFOLD", "0")
        passes = m.forward_tokens(x, upto=k).float()
        assert calls["fold"] == 2 * k and calls["ln"] == 2 * k, calls
    rel = ((folded - passes).pow(2).mean().sqrt() / passes.pow(2).mean().sqrt()).item()
    util.record_margin("vit_h_folded_vs_passes", rel_4=rel)
    assert rel <= 1.5 * E_BLOCK * (k + 1) ** 0.5, rel


def test_preprocess_matches_oracle():
    from sam6d_amd.sam.image_encoder import preprocess
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 768, 1024, generator=g) * 255
    ref = torch.stack([osam.preprocess(x[i]) for i in range(2)])
    out32 = preprocess(x.cuda(), out_dtype=torch.float32).cpu()
    assert out32.shape == (2, 3, 1024, 1024) and (out32 - ref).abs().max() < 1e-5
    out16 = preprocess(x.cuda(), out_dtype=torch.bfloat16).float().cpu()
    assert (out16 - ref).abs().max() < 2e-2 and (out16[:, :, 768:] == 0).all()


def test_layout_kernels_bit_exact_vs_the_library_statements():
    """s6d_patchify_b16 / s6d_im2col3x3_b16 move two-byte elements: the results must equal PatchEmbed's view / permute / reshape and
    the nine shifted views of the zero-padded map (what the neck's 3x3 convolution read before) bit for bit."""
    import torch.nn.functional as F

    from sam6d_amd import ops
    g = torch.Generator().manual_seed(1)
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(2, 3, 64, 96, generator=g).to(dt)
        p = 16
        ref = x.view(2, 3, 4, p, 6, p).permute(0, 2, 4, 1, 3, 5).reshape(2, 4, 6, 3 * p * p)
        assert torch.equal(ops.patchify(x.cuda(), p).cpu(), ref)
        y = torch.randn(2, 5, 7, 16, generator=g).to(dt)
        yp = F.pad(y, (0, 0, 1, 1, 1, 1))
        ref = torch.cat([yp[:, dy:dy + 5, dx:dx + 7, :] for dy in range(3) for dx in range(3)], dim=-1)
        assert torch.equal(ops.im2col3x3(y.cuda()).cpu(), ref)
    x8 = torch.randn(1, 2, 8, 8, generator=g).bfloat16()             # p = 8, one patch per channel row
    assert torch.equal(ops.patchify(x8.cuda(), 8).cpu(), x8.view(1, 2, 1, 8, 1, 8).permute(0, 2, 4, 1, 3, 5).reshape(1, 1, 1, 128))
    with pytest.raises(RuntimeError):
        ops.im2col3x3(torch.zeros(1, 2, 2, 12, dtype=torch.bfloat16).cuda())         # C % 8 != 0


@pytest.mark.parametrize("index", [0, 7])
def test_vit_h_block_bf16_hip_path_vs_oracle_block(index):
    """ONE ViT-H block through the bf16 HIP path (fused residual + LayerNorm, hand-written GEMMs with bias / GELU epilogues, the
    window (index 0) or 64 x 64 global (index 7) attention kernel) against oracle.sam.block -- the statement of Block.forward that
    tests/test_oracle_golden.py pins to the reference -- evaluated in fp32 on the SAME bf16-rounded input, GEMM weights and
    position tables.  What is left is the path's own rounding: activations are stored in bf16 between its six stages (2^-9
    relative each), sums are fp32.  Bound: rms error <= 4e-3 of the output rms (measured 2.7e-3), no element off by more than 2^-6 of the
    largest output (measured 5.4e-3) -- an element-level statement the whole-encoder correlation test cannot make."""
    from sam6d_amd.sam.image_encoder import build_vit_h
    m = seeded.load_seeded(build_vit_h().eval(), 3)
    blk = m.blocks[index]
    g = torch.Generator().manual_seed(40 + index)
    x = (0.5 * torch.randn(1, 64, 64, 1280, generator=g)).to(torch.bfloat16)
    name = f"blocks.{index}"
    W = {}
    for k, v in blk.state_dict().items():
        v = v.float()
        if k.endswith(("qkv.weight", "proj.weight", "lin1.weight", "lin2.weight", "rel_pos_h", "rel_pos_w")):
            v = v.to(torch.bfloat16).float()
        W[f"{name}.{k}"] = v
    with torch.no_grad():
        ref = osam.block(W, name, x.float(), 16, blk.window_size)
        m.blocks = torch.nn.ModuleList([blk])
        m = m.cuda()
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            out = m._blocks_fused(x.cuda(), None).float().cpu()
    err = (out - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    assert err.pow(2).mean().sqrt() <= 4e-3 * rms and err.max() <= 2.0 ** -6 * ref.abs().max(), \
        (float(err.pow(2).mean().sqrt() / rms), float(err.max() / ref.abs().max()))
