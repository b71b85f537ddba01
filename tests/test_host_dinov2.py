"""Host-logic tests of the drop-in DINOv2 descriptor model on CPU: state_dict surface, the fp32 library-op path of
the ViT against the reference golden, and the crop geometry table (crop_params) replayed with the kernel's index rule
in numpy against the reference's crops -- no GPU, no HIP library calls."""
import ast

import numpy as np
import pytest
import torch

from oracle import dinov2 as odino
from sam6d_amd.ism import dinov2 as pd
from sam6d_amd.utils import seeded, synth
from tests import util


def _case():
    g = util.golden("dinov2.npz")
    c = ast.literal_eval(str(g["case"]))
    return g, c, synth.dinov2_inputs(P=c["P"], seed=c["input_seed"])


def _mini():
    c = odino.MINI
    return pd.DinoVisionTransformer(img_size=c["img_size"], patch_size=c["patch"], embed_dim=c["dim"], depth=c["depth"],
                                    num_heads=c["heads"], mlp_ratio=4, init_values=1.0, block_chunks=0).eval()


def test_state_dict_surface_mini_and_vit_l():
    g, _, _ = _case()
    assert sorted(_mini().state_dict().keys()) == [str(k) for k in g["mini_keys"]]
    with torch.device("meta"):
        m = pd._make_dinov2_model(arch_name="vit_large")
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(v) for k, v in util.shapes_from_golden(g, "l_keys", "l_shapes").items()}


def test_mini_vit_library_path_matches_reference_golden():
    g, c, _ = _case()
    m = _mini()
    seeded.load_seeded(m, c["weight_seed"])
    rgbs, masks = torch.from_numpy(g["mini_rgbs"]), torch.from_numpy(g["mini_masks"])
    o = pd.CustomDINOv2.__new__(pd.CustomDINOv2)
    torch.nn.Module.__init__(o)
    o.model, o.patch_size, o.validpatch_thresh, o.chunk_size = m, 14, 0.5, 3
    cls, patch = o.compute_cls_and_patch_features(rgbs, masks)
    np.testing.assert_allclose(cls.numpy(), g["mini_cls"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(patch.numpy(), g["mini_patch"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o.compute_masked_patch_feature(rgbs, masks).numpy(), g["mini_patch"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o.compute_features(rgbs, "x_norm_clstoken").numpy(), g["mini_cls"], rtol=1e-4, atol=1e-5)


def _replay(rec, image_u8, masks, T):
    """numpy statement of s6d_crop_resize_pad_f32's index rule driven by the crop_params table."""
    def src(dst, n_in, inv):
        return np.minimum(np.floor(dst.astype(np.float32) * np.float32(inv)).astype(np.int64), n_in - 1)
    rgb = odino.rgb_normalize(image_u8).numpy()
    P = rec.shape[0]
    out = np.zeros((P, 3, T, T), np.float32)
    om = np.zeros((P, T, T), np.float32)
    o = np.arange(T)
    for p in range(P):
        x1, y1, h, w, h1, w1, top, left, S2 = [int(v) for v in rec[p, :9]]
        inv1, inv2 = rec[p, 9:11].view(np.float32)
        py, px = src(o, S2, inv2) - top, src(o, S2, inv2) - left
        vy, vx = (py >= 0) & (py < h1), (px >= 0) & (px < w1)
        sy = y1 + src(np.clip(py, 0, h1 - 1), h, inv1)
        sx = x1 + src(np.clip(px, 0, w1 - 1), w, inv1)
        valid = vy[:, None] & vx[None, :]
        mk = masks[p][sy[:, None], sx[None, :]] * valid
        om[p] = mk
        out[p] = rgb[:, sy[:, None], sx[None, :]] * mk * valid
    return out, om


@pytest.mark.parametrize("target,key", [(56, "mini")])
def test_crop_params_replay_matches_reference_crops(target, key):
    g, _, inp = _case()
    rec = pd.crop_params(inp["boxes"].numpy(), target)
    rgbs, masks = _replay(rec, inp["image"], inp["masks"].numpy(), target)
    np.testing.assert_array_equal(masks, g[key + "_masks"])
    np.testing.assert_array_equal(rgbs, g[key + "_rgbs"])


def test_crop_params_224_digest_and_reference_failure_mode():
    g, _, inp = _case()
    rec = pd.crop_params(inp["boxes"].numpy(), 224)
    rgbs, masks = _replay(rec, inp["image"], inp["masks"].numpy(), 224)
    util.assert_digest_close(torch.from_numpy(rgbs), g["l_rgbs_sum"], g["l_rgbs_smp"], 1009, 0, 0, "224 crops")
    util.assert_digest_close(torch.from_numpy(masks), g["l_masks_sum"], g["l_masks_smp"], 1009, 0, 0, "224 masks")
    # a 99 x 99 crop at target 56: the reference's second resize floors to 55 and its torch.stack raises; same here
    with pytest.raises(RuntimeError, match="equal size"):
        pd.crop_params(np.array([[10, 10, 109, 109], [0, 0, 50, 20]]), 56)
    with pytest.raises(RuntimeError):
        pd.crop_params(np.array([[10, 10, 10, 40]]), 224)
    assert pd.crop_params(np.zeros((0, 4), np.int64), 224).shape == (0, 12)
    # crop_valid flags exactly the boxes on which crop_params raises
    probe = np.array([[10, 10, 109, 109], [0, 0, 50, 20], [10, 10, 10, 40], [5, 5, 105, 105], [0, 0, 600, 2]])
    valid = pd.crop_valid(probe, 56)
    assert valid.tolist() == [False, True, False, True, False]
    for bx, ok in zip(probe, valid):
        if ok:
            pd.crop_params(bx[None], 56)
        else:
            with pytest.raises((RuntimeError, AssertionError)):
                pd.crop_params(bx[None], 56)


def test_crop_params_replay_random_boxes_vs_oracle():
    """~290 random boxes (every aspect ratio; sides 2..full frame) against the reference algorithm: pins the scale
    arithmetic (reciprocal * target), the floor'ed sizes and the index rule incl. the out == in case."""
    g = torch.Generator().manual_seed(3)
    H, W, P = 480, 640, 300
    img = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy()
    boxes = synth.random_boxes(P, H, W, g)
    masks = (torch.rand(len(boxes), H, W, generator=g) > 0.3).float()
    rec = pd.crop_params(boxes.numpy(), 224)
    rgbs, pm = _replay(rec, img, masks.numpy(), 224)
    np.testing.assert_array_equal(pm, odino.process_masks_proposals(masks, boxes, 224).numpy())
    np.testing.assert_array_equal(rgbs, odino.process_rgb_proposals(img, masks, boxes, 224).numpy())


def test_product_ops_refuse_cpu_tensors():
    _, _, inp = _case()
    o = pd.CustomDINOv2.__new__(pd.CustomDINOv2)
    torch.nn.Module.__init__(o)
    o.proposal_size = 56
    with pytest.raises(RuntimeError):
        o.process_rgb_proposals(inp["image"], inp["masks"], inp["boxes"])
