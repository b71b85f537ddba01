"""Frame resize of the SAM path (sam6d_amd/sam/transforms.py) against the reference's ResizeLongestSide run unmodified
(golden: oracle/gen_golden.py sam_transforms) and against Pillow itself: the uint8 pixels are bit-identical."""
import numpy as np
import pytest
import torch

from sam6d_amd.sam.transforms import ResizeLongestSide, pil_bilinear_coeffs, pil_bilinear_resize_u8
from tests import util

TAGS = (("vga", (60, 80)), ("tless", (54, 72)), ("itodd", (96, 128)), ("tall", (75, 31)))


def test_resize_longest_side_matches_the_reference():
    g = util.golden("sam_transforms.npz")
    for tag, hw in TAGS:
        t = ResizeLongestSide(int(g[tag + "_L"]))
        img = g[tag + "_img"]
        out = t.apply_image(img)
        assert isinstance(out, np.ndarray) and out.dtype == np.uint8
        np.testing.assert_array_equal(out, g[tag + "_out"])
        np.testing.assert_array_equal(t.apply_image(torch.from_numpy(img)).numpy(), g[tag + "_out"])   # tensor in, tensor out
        pts, boxes = g[tag + "_pts"], g[tag + "_boxes"]
        keep = pts.copy()
        np.testing.assert_array_equal(t.apply_coords(pts, hw), g[tag + "_pts_out"])
        np.testing.assert_array_equal(pts, keep)                                                          # input untouched
        np.testing.assert_array_equal(t.apply_boxes(boxes, hw), g[tag + "_boxes_out"])
        np.testing.assert_array_equal(t.apply_coords_torch(torch.from_numpy(pts), hw).numpy(), g[tag + "_pts_out_t"])
        np.testing.assert_array_equal(t.apply_boxes_torch(torch.from_numpy(boxes), hw).numpy(), g[tag + "_boxes_out_t"])
    assert ResizeLongestSide.get_preprocess_shape(480, 640, 1024) == (768, 1024)
    assert ResizeLongestSide.get_preprocess_shape(540, 720, 1024) == (768, 1024)
    assert ResizeLongestSide.get_preprocess_shape(1, 1000, 1024) == (1, 1024)


def test_resampler_is_bit_identical_with_pillow_at_frame_sizes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    cases = [((480, 640), (768, 1024)), ((540, 720), (768, 1024)), ((960, 1280), (768, 1024)), ((37, 53), (101, 7)),
             ((100, 333), (30, 1024)), ((64, 64), (64, 128)), ((50, 70), (20, 70)), ((3, 5), (9, 2)), ((17, 1), (5, 3)),
             ((1, 1), (4, 4)), ((33, 44), (33, 44))]
    for (H, W), (h, w) in cases:
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        a[::3] = 255                                                          # saturated rows: the clip after each pass
        ref = np.array(Image.fromarray(a).resize((w, h), Image.BILINEAR))
        np.testing.assert_array_equal(pil_bilinear_resize_u8(torch.from_numpy(a), (h, w)).numpy(), ref, err_msg=str(((H, W), (h, w))))
        ref1 = np.array(Image.fromarray(a[:, :, 0]).resize((w, h), Image.BILINEAR))
        np.testing.assert_array_equal(pil_bilinear_resize_u8(torch.from_numpy(a[:, :, 0].copy()), (h, w)).numpy(), ref1)


def test_coefficient_tables_and_argument_checks():
    xmin, kk = pil_bilinear_coeffs(640, 1024)
    assert kk.shape == (1024, 3) and xmin.shape == (1024,) and kk.dtype == np.int32
    assert (np.abs(kk.sum(1) - (1 << 22)) <= 2).all()                         # weights sum to one within the quantisation
    xmin, kk = pil_bilinear_coeffs(1280, 1024)
    assert kk.shape == (1024, 5)                                              # shrinking widens the window
    with pytest.raises(ValueError):
        pil_bilinear_resize_u8(torch.zeros(4, 4, 3), (8, 8))                  # not uint8
    with pytest.raises(ValueError):
        pil_bilinear_resize_u8(torch.zeros(4, 4, 3, dtype=torch.uint8), (0, 8))
    same = torch.arange(48, dtype=torch.uint8).reshape(4, 4, 3)
    out = pil_bilinear_resize_u8(same, (4, 4))
    assert torch.equal(out, same) and out.data_ptr() != same.data_ptr()       # identity resize returns a copy
