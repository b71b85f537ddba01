"""GPU parity of the DINOv2 descriptor path: fused proposal crops (bit-exact vs the oracle
and the reference goldens) and the ViT descriptors (fp32 tight, bf16 within round-off) -- through the C ABI."""
import ast
import types

import numpy as np
import pytest
import torch

from oracle import dinov2 as odino
from sam6d_amd.ism import dinov2 as pd
from sam6d_amd.utils import seeded, synth
from tests import util

pytestmark = pytest.mark.gpu


def _case():
    g = util.golden("dinov2.npz")
    c = ast.literal_eval(str(g["case"]))
    return g, c, synth.dinov2_inputs(P=c["P"], seed=c["input_seed"])


def _custom(model, target, chunk=3):
    o = pd.CustomDINOv2.__new__(pd.CustomDINOv2)
    torch.nn.Module.__init__(o)
    o.model, o.patch_size, o.validpatch_thresh, o.chunk_size, o.proposal_size = model, 14, 0.5, chunk, target
    o.token_name = "x_norm_clstoken"
    return o


def _mini():
    c = odino.MINI
    return pd.DinoVisionTransformer(img_size=c["img_size"], patch_size=c["patch"], embed_dim=c["dim"], depth=c["depth"],
                                    num_heads=c["heads"], mlp_ratio=4, init_values=1.0, block_chunks=0).eval()


@pytest.mark.parametrize("target", [56, 224])
def test_crops_bit_exact_vs_oracle_and_golden(target):
    g, _, inp = _case()
    o = _custom(None, target)
    masks, boxes = inp["masks"].cuda(), inp["boxes"].cuda()
    rgbs = o.process_rgb_proposals(inp["image"], masks, boxes)
    pm = o.process_masks_proposals(masks.clone(), boxes)
    ref_rgb = odino.process_rgb_proposals(inp["image"], inp["masks"], inp["boxes"], target)
    ref_m = odino.process_masks_proposals(inp["masks"], inp["boxes"], target)
    assert torch.equal(rgbs.cpu(), ref_rgb) and torch.equal(pm.cpu(), ref_m)
    both = o._crops(inp["image"], masks, boxes, True, True)
    assert torch.equal(both[0], rgbs) and torch.equal(both[1], pm)
    if target == 56:
        np.testing.assert_array_equal(rgbs.cpu().numpy(), g["mini_rgbs"])
        np.testing.assert_array_equal(pm.cpu().numpy(), g["mini_masks"])
    else:
        util.assert_digest_close(rgbs, g["l_rgbs_sum"], g["l_rgbs_smp"], 1009, 0, 0, "224 crops")


def test_crops_random_boxes_bit_exact_vs_oracle():
    """~150 random boxes on a 480x640 frame (all aspect ratios, sizes 2..full) against the reference algorithm."""
    g = torch.Generator().manual_seed(5)
    H, W, P = 480, 640, 160
    img = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy()
    boxes = synth.random_boxes(P, H, W, g)
    masks = (torch.rand(len(boxes), H, W, generator=g) > 0.3).float()
    o = _custom(None, 224)
    rgbs, pm = o._crops(img, masks.cuda(), boxes.cuda(), True, True)
    assert torch.equal(rgbs.cpu(), odino.process_rgb_proposals(img, masks, boxes, 224))
    assert torch.equal(pm.cpu(), odino.process_masks_proposals(masks, boxes, 224))


def test_crop_edge_cases():
    _, _, inp = _case()
    o = _custom(None, 56)
    e = o._crops(inp["image"], inp["masks"][:0].cuda(), inp["boxes"][:0].cuda(), True, True)
    assert e[0].shape == (0, 3, 56, 56) and e[1].shape == (0, 56, 56)
    with pytest.raises(RuntimeError, match="equal size"):
        o._crops(inp["image"], inp["masks"][:1].cuda(), torch.tensor([[10, 10, 109, 109]]).cuda(), True, True)
    with pytest.raises(RuntimeError):
        o.process_rgb_proposals(inp["image"], inp["masks"], inp["boxes"])          # CPU tensors are refused


def test_mini_descriptors_fp32_and_bf16_vs_reference_golden(monkeypatch):
    g, c, inp = _case()
    m = seeded.load_seeded(_mini(), c["weight_seed"]).cuda()
    o = _custom(m, c["mini_target"])
    props = types.SimpleNamespace(masks=inp["masks"].cuda(), boxes=inp["boxes"].cuda())
    monkeypatch.setenv("S6D_DINO_DTYPE", "fp32")
    cls, patch = o.forward(inp["image"], props)
    np.testing.assert_allclose(cls.cpu().numpy(), g["mini_cls"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(patch.cpu().numpy(), g["mini_patch"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(o.forward_cls_token(inp["image"], props).cpu().numpy(), g["mini_cls"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(o.forward_patch_tokens(inp["image"], props).cpu().numpy(), g["mini_patch"], rtol=1e-3,
                               atol=2e-4)
    monkeypatch.setenv("S6D_DINO_DTYPE", "bf16")                               # fused kernels (head dim 64)
    assert m._fusable(torch.zeros(1, 2, 128, device="cuda", dtype=torch.bfloat16))
    cls, patch = o.forward(inp["image"], props)
    cls, patch = cls.cpu().numpy(), patch.cpu().numpy()
    assert np.abs(cls - g["mini_cls"]).mean() < 2e-2 and np.corrcoef(cls.ravel(), g["mini_cls"].ravel())[0, 1] > 0.999
    # masked rows are exactly zero in both; kept rows are unit vectors whose cosine to the reference is ~1
    zero = np.abs(g["mini_patch"]).sum(-1) == 0
    assert (np.abs(patch).sum(-1)[zero] == 0).all()
    cos = (patch * g["mini_patch"]).sum(-1)[~zero]
    assert cos.min() > 0.999, cos.min()


E_BLOCK_L = 4e-3     # per-block rms rounding of the bf16 path relative to the residual stream's rms (as the ViT-H's, tests/test_gpu_sam.py)


def test_vit_l14_bf16_error_growth_model(monkeypatch):
    """ViT-L/14 in the dtype the descriptor stage runs (bf16, folded block loop) held to the ERROR MODEL the ViT-H is held to
     instead of a cosine: every block adds an independent rms rounding of at most E_BLOCK_L of the residual
    stream's rms, so after k blocks the stream sits within E_BLOCK_L * sqrt(k + 1) of the fp32 path's ("+1": the bf16 patch
    embedding + positional term).  Checked at k = 1, 2, 4, 8, 16, 24 against the SAME model in fp32 on the device (pinned to the
    reference golden below), on 8 crops of the golden's frame."""
    import torch.nn as nn
    g, c, inp = _case()
    m = seeded.load_seeded(pd._make_dinov2_model(arch_name="vit_large").eval(), c["weight_seed"]).cuda()
    o = _custom(m, 224, chunk=128)
    rgbs, _ = o._crops(inp["image"], inp["masks"].cuda(), inp["boxes"].cuda(), True, True)
    full = m.blocks
    rel = {}
    try:
        for k in (1, 2, 4, 8, 16, 24):
            m.blocks = nn.ModuleList(list(full)[:k])
            with torch.no_grad():
                monkeypatch.setenv("S6D_DINO_DTYPE", "fp32")
                t32 = m.forward_features(rgbs)["x_prenorm"].float()
                monkeypatch.setenv("S6D_DINO_DTYPE", "bf16")
                t16 = m.forward_features(rgbs)["x_prenorm"].float()
            rel[k] = ((t16 - t32).pow(2).mean().sqrt() / t32.pow(2).mean().sqrt()).item()
    finally:
        m.blocks = full
    util.record_margin("vit_l14_bf16_error_growth", **{f"rel_{k}": v for k, v in rel.items()})
    for k, v in rel.items():
        assert v <= E_BLOCK_L * (k + 1) ** 0.5, rel


def test_vit_l14_bf16_vs_reference_golden(monkeypatch):
    """Released configuration (ViT-L/14, 224 crops, pos-embed interpolated 37x37 -> 16x16), fused bf16 pipeline against the
    reference's fp32 descriptors: the final LayerNorm renormalises the stream, so the cls descriptors sit within the error model's
    E_BLOCK_L * sqrt(25) of the reference's (rms of the difference relative to the reference's rms, per descriptor), the sampled
    patch descriptors (unit vectors) likewise."""
    g, c, inp = _case()
    monkeypatch.setenv("S6D_DINO_DTYPE", "bf16")
    m = seeded.load_seeded(pd._make_dinov2_model(arch_name="vit_large").eval(), c["weight_seed"]).cuda()
    o = _custom(m, 224, chunk=128)
    n = c["n_full"]
    props = types.SimpleNamespace(masks=inp["masks"][:n].cuda(), boxes=inp["boxes"][:n].cuda())
    cls, patch = o.forward(inp["image"], props)
    cls = cls.cpu().numpy()
    bound = E_BLOCK_L * 25 ** 0.5
    rel = np.linalg.norm(cls - g["l_cls"], axis=-1) / np.linalg.norm(g["l_cls"], axis=-1)
    smp = patch.cpu().reshape(-1)[::53].numpy()
    rel_p = np.sqrt(((smp - g["l_patch_smp"]) ** 2).mean() / (g["l_patch_smp"] ** 2).mean())
    util.record_margin("vit_l14_bf16_vs_reference", cls_rel_max=float(rel.max()), patch_rel_rms=float(rel_p), bound=bound)
    assert rel.max() <= bound, rel
    assert rel_p <= bound, rel_p
