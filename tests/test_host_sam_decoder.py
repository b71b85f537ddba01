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
    with torch.device("meta"):
        m = md.build_sam_decoder()                                   # the build_sam._build_sam construction
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
        {k: tuple(v) for k, v in util.shapes_from_golden(g, "sam_keys", "sam_shapes").items()}


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


def _bil_np(img, oh, ow, ih, iw):
    """numpy statement of the arithmetic s6d_sam_mask_post_f32 uses for one bilinear stage (fma emulated in float64:
    the product of two float32 is exact there)."""
    f = np.float32

    def taps(out_n, in_n):
        d = np.arange(out_n, dtype=np.float32)
        s = np.maximum(f(f(in_n) / f(out_n)) * (d + f(0.5)) - f(0.5), f(0))
        i0 = s.astype(np.int64)
        i1 = i0 + (i0 < in_n - 1)
        w1 = (s - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (f(1) - w1).astype(np.float32), w1

    def mix(w0, a, w1, b):
        return (w0.astype(np.float64) * a.astype(np.float64) + (w1 * b).astype(np.float32).astype(np.float64)).astype(np.float32)
    y0, y1, wy0, wy1 = taps(oh, ih)
    x0, x1, wx0, wx1 = taps(ow, iw)
    WX0, WX1 = np.broadcast_to(wx0[None, :], (oh, ow)), np.broadcast_to(wx1[None, :], (oh, ow))
    WY0, WY1 = np.broadcast_to(wy0[:, None], (oh, ow)), np.broadcast_to(wy1[:, None], (oh, ow))
    t0 = mix(WX0, img[np.ix_(y0, x0)], WX1, img[np.ix_(y0, x1)])
    t1 = mix(WX0, img[np.ix_(y1, x0)], WX1, img[np.ix_(y1, x1)])
    return mix(WY0, t0, WY1, t1)


def test_mask_post_arithmetic_replay_matches_oracle_and_reference_golden():
    """The kernel's bilinear form (ATen CPU's, operation for operation) reproduces the oracle's upscaled logits bit for
    bit; stability scores / boxes / areas of the thresholded masks equal the reference amg.py outputs."""
    g, c, _, _ = case("mini")
    low = synth.sam_lowres_logits(c["post_B"], 3, 256, c["post_seed"])
    (ih, iw), (H, W) = c["post_input_size"], c["post_orig"]
    full = osd.postprocess_masks(low, 1024, (ih, iw), (H, W)).flatten(0, 1).numpy()
    for m in (1, 4, 8):
        mine = _bil_np(_bil_np(low.flatten(0, 1)[m].numpy(), 1024, 1024, 256, 256)[:ih, :iw], H, W, ih, iw)
        np.testing.assert_array_equal(mine, full[m])
    mb, st, boxes = osd.mask_postprocess(low, 1024, (ih, iw), (H, W))
    np.testing.assert_array_equal(st.numpy(), g["post_stability"])          # NaN == NaN under assert_array_equal
    np.testing.assert_array_equal(boxes.numpy(), g["post_boxes"])
    np.testing.assert_array_equal(mb.flatten(1).sum(1).numpy(), g["post_area"])
    np.testing.assert_array_equal(np.packbits(mb.numpy().reshape(mb.shape[0], -1)[:, ::7], axis=1), g["post_bits"])


def test_generator_bookkeeping_matches_reference():
    """Point grid, ResizeLongestSide shape and prompt coordinates (utils/amg.py:179-186, utils/transforms.py:33-43,95-102)."""
    from sam6d_amd.sam import amg
    g = util.golden("sam_decoder.npz")
    grid = amg.build_point_grid(32)
    np.testing.assert_array_equal(grid, g["grid32"])
    for (h, w), ref in zip(g["pre_sizes"], g["pre_shapes"]):
        assert amg.preprocess_shape(int(h), int(w), 1024) == tuple(ref)
    ih, iw = amg.preprocess_shape(480, 640, 1024)
    pts = (grid * [[640, 480]]) * [[iw / 640, ih / 480]]                 # what generate_proposals feeds the prompt encoder
    np.testing.assert_array_equal(pts, g["coords_480x640"])
