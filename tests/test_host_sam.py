"""Host-logic tests of the drop-in SAM encoder on CPU (library-op path, fp32)."""
from functools import partial

import numpy as np
import torch

from oracle import sam as osam
from sam6d_amd.utils import seeded, synth
from tests import util


def _mini():
    from sam6d_amd.sam.image_encoder import ImageEncoderViT
    c = osam.MINI
    return ImageEncoderViT(depth=c["depth"], embed_dim=c["dim"], img_size=c["img_size"], mlp_ratio=4,
                           norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=c["heads"], patch_size=16,
                           qkv_bias=True, use_rel_pos=True, global_attn_indexes=c["global_idx"],
                           window_size=c["window"], out_chans=c["out_chans"]).eval()


def test_mini_encoder_matches_reference_golden_and_keys():
    g = util.golden("sam_enc.npz")
    m = _mini()
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["mini_keys"]]
    seeded.load_seeded(m, 3)
    with torch.no_grad():
        y = m(synth.sam_input(1, 5, osam.MINI["img_size"]))
    np.testing.assert_allclose(y.numpy(), g["mini_out"], rtol=1e-4, atol=2e-5)


def test_vit_h_state_dict_surface():
    from sam6d_amd.sam.image_encoder import build_vit_h
    g = util.golden("sam_enc.npz")
    with torch.device("meta"):
        m = build_vit_h()
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(v) for k, v in util.shapes_from_golden(g, "h_keys", "h_shapes").items()}
    assert m.img_size == 1024


def test_mlp_row_chunk_knob_is_the_same_function(monkeypatch):
    """S6D_SAM_MLP_ROWS (experiment knob, off by default) only changes how many rows go through lin1 -> GELU -> lin2 at a time."""
    import torch

    from sam6d_amd.sam.image_encoder import MLPBlock
    torch.manual_seed(0)
    m = MLPBlock(32, 128).eval()
    x = torch.randn(2, 5, 7, 32)
    with torch.no_grad():
        want = m(x)
        for rows in ("16", "33", "70", "1000"):
            monkeypatch.setenv("S6D_SAM_MLP_ROWS", rows)
            got = m(x)
            assert got.shape == want.shape and torch.allclose(got, want, atol=1e-6), rows

