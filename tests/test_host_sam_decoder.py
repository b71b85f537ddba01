"""Host-logic tests of the drop-in SAM prompt encoder + mask decoder on CPU (fp32): state_dict surface and both
execution paths (the reference's op sequence, and the restructured shared-image-token path) against the reference
golden."""
import ast

import numpy as np
import torch

from oracle import sam_decoder as osd
from sam6d_amd.sam import mask_decoder as md
from sam6d_amd.utils import seeded, synth
from tests import util


def build(cfg):
    m = torch.nn.Module()
    m.prompt_encoder = md.PromptEncoder(embed_dim=cfg["dim"], image_embedding_size=(cfg["emb"],) * 2,
                                        input_image_size=(cfg["img"],) * 2, mask_in_chans=16)
    m.mask_decoder = md.MaskDecoder(num_multimask_outputs=cfg["n_multi"],
                                    transformer=md.TwoWayTransformer(depth=cfg["depth"], embedding_dim=cfg["dim"],
                                                                     mlp_dim=cfg["mlp"], num_heads=cfg["heads"]),
                                    transformer_dim=cfg["dim"], iou_head_depth=cfg["iou_depth"],
                                    iou_head_hidden_dim=cfg["iou_hidden"])
    return m.eval()


def case(name):
    g = util.golden("sam_decoder.npz")
    c = ast.literal_eval(str(g["case"]))
    cfg = osd.MINI if name == "mini" else osd.SAM
    inp = synth.sam_decoder_inputs(cfg, c["n_mini"] if name == "mini" else c["n_full"], c["input_seed"])
    return g, c, cfg, inp


def run(m, emb, points=None, boxes=None, multi=True, force_lib=False):
    s, d = m.prompt_encoder(points=points, boxes=boxes, masks=None)
    if force_lib:
        d = d.contiguous()                      # a materialised dense embedding takes the reference's op sequence
    mk, iou = m.mask_decoder(image_embeddings=emb, image_pe=m.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=s,
                             dense_prompt_embeddings=d, multimask_output=multi)
    return s, mk, iou


def test_state_dict_surface():
    g = util.golden("sam_decoder.npz")
    for name, cfg in (("mini", osd.MINI), ("sam", osd.SAM)):
        with torch.device("meta"):
            m = build(cfg)
        mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert mine == {k: tuple(v) for k, v in util.shapes_from_golden(g, name + "_keys", name + "_shapes").items()}


def test_mini_both_paths_match_reference_golden():
    g, c, cfg, inp = case("mini")
    m = seeded.load_seeded(build(cfg), c["weight_seed"])
    with torch.no_grad():
        np.testing.assert_allclose(m.prompt_encoder.get_dense_pe().numpy(), g["mini_dense_pe"], rtol=1e-5, atol=1e-6)
        for force_lib in (False, True):
            for tag, kw in (("", dict(points=(inp["points"], inp["labels"]))),
                            ("2", dict(points=(inp["points2"], inp["labels2"]), multi=False)),
                            ("_box", dict(boxes=inp["boxes"]))):
                s, mk, iou = run(m, inp["emb"], force_lib=force_lib, **kw)
                np.testing.assert_allclose(s.numpy(), g["mini_sparse" + tag], rtol=1e-5, atol=1e-6)
                np.testing.assert_allclose(mk.numpy(), g["mini_masks" + tag], rtol=1e-4, atol=2e-5)
                np.testing.assert_allclose(iou.numpy(), g["mini_iou" + tag], rtol=1e-4, atol=2e-5)


def test_released_config_shared_path_matches_reference_golden():
    g, c, cfg, inp = case("sam")
    m = seeded.load_seeded(build(cfg), c["weight_seed"])
    with torch.no_grad():
        _, mk, iou = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
    np.testing.assert_allclose(iou.numpy(), g["sam_iou"], rtol=1e-4, atol=2e-5)
    util.assert_digest_close(mk, g["sam_masks_sum"], g["sam_masks_smp"], 211, 1e-4, 2e-5, "low-res mask logits")
