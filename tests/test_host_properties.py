"""Property tests (hypothesis) of the byte / integer host paths at random shapes: the frame resampler against Pillow, the
run-length encoder against a per-pixel loop and its decoder, the square-box rule against a scalar restatement."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from sam6d_amd.ism import handoff
from sam6d_amd.pem import preprocess
from sam6d_amd.sam.transforms import pil_bilinear_resize_u8

Image = pytest.importorskip("PIL.Image")


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 70), st.integers(1, 70), st.integers(1, 90), st.integers(1, 90), st.integers(0, 2 ** 31 - 1), st.sampled_from([1, 3]))
def test_resampler_equals_pillow(H, W, h, w, seed, C):
    a = np.random.default_rng(seed).integers(0, 256, (H, W, C), dtype=np.uint8)
    ref = np.array(Image.fromarray(a if C == 3 else a[:, :, 0]).resize((w, h), Image.BILINEAR)).reshape(h, w, C)
    np.testing.assert_array_equal(pil_bilinear_resize_u8(torch.from_numpy(a), (h, w)).numpy(), ref)


def _rle_loop(mask):
    """mask_to_rle of the reference (model/utils.py:24-43) as a plain loop over the column-major pixels."""
    counts, last, run = [], 0, 0
    for v in mask.T.reshape(-1).tolist():
        if v != last:
            counts.append(run)
            run, last = 0, v
        run += 1
    counts.append(run)
    return {"counts": counts, "size": list(mask.shape)}


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 5), st.integers(1, 23), st.integers(1, 23), st.floats(0.0, 1.0), st.integers(0, 2 ** 31 - 1))
def test_rle_equals_pixel_loop_and_round_trips(n, H, W, p, seed):
    m = np.random.default_rng(seed).random((n, H, W)) < p
    rles = handoff.masks_to_rle(torch.from_numpy(m))
    for i in range(n):
        assert rles[i] == _rle_loop(m[i].astype(np.uint8))
        assert np.array_equal(handoff.rle_to_mask(rles[i]), m[i])


def _bbox_scalar(mask):
    """get_bbox of the reference (Pose_Estimation_Model/utils/data_utils.py:126-160) for one mask, as scalar arithmetic
    (checked equal to the reference function itself on 3000 random masks when this test was written)."""
    H, W = mask.shape
    rows, cols = np.where(mask.any(1))[0], np.where(mask.any(0))[0]
    rmin, rmax, cmin, cmax = rows[0], rows[-1] + 1, cols[0], cols[-1] + 1
    b = min(max(rmax - rmin, cmax - cmin), min(H, W))
    cy, cx = (rmin + rmax) // 2, (cmin + cmax) // 2
    rmin, rmax, cmin, cmax = cy - b // 2, cy + b // 2, cx - b // 2, cx + b // 2
    if rmin < 0:
        rmax, rmin = rmax - rmin, 0
    if cmin < 0:
        cmax, cmin = cmax - cmin, 0
    if rmax > H:
        rmin, rmax = rmin - (rmax - H), H
    if cmax > W:
        cmin, cmax = cmin - (cmax - W), W
    return [rmin, rmax, cmin, cmax]


@settings(max_examples=80, deadline=None)
@given(st.integers(2, 40), st.integers(2, 40), st.integers(0, 2 ** 31 - 1), st.floats(0.02, 0.9))
def test_square_boxes_equal_the_scalar_rule(H, W, seed, p):
    rng = np.random.default_rng(seed)
    m = rng.random((3, H, W)) < p
    m[:, rng.integers(0, H), rng.integers(0, W)] = True              # at least one pixel each
    got = preprocess.square_boxes(torch.from_numpy(m)).tolist()
    assert got == [[int(v) for v in _bbox_scalar(m[i])] for i in range(3)]
