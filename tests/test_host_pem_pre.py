"""PEM per-detection pre-processing: the oracle's geometry helpers against the reference's
(golden), and the batched device implementation (plain tensor ops: it also runs on the CPU) against the oracle's
per-detection loop on a frame with square / tall / wide / tiny / near-full / random proposals."""
import numpy as np
import torch

from oracle import pem_pre as opre
from sam6d_amd.pem import preprocess as pre
from sam6d_amd.utils import synth
from tests import util


def test_oracle_helpers_match_reference_golden():
    g = util.golden("pem_pre.npz")
    inp = synth.pem_pre_inputs(P=8, seed=3)
    masks, depth, K = inp["masks"].numpy(), inp["depth"].numpy(), inp["K"].numpy()
    bbox = np.array([opre.get_bbox(np.logical_and(m, depth > 0)) for m in masks])
    np.testing.assert_array_equal(bbox, g["bbox"])
    cloud = opre.point_cloud(depth, K)
    util.assert_digest_close(torch.from_numpy(cloud), g["cloud_sum"], g["cloud_smp"], 499, 1e-6, 1e-7, "back-projection")
    y1, y2, x1, x2 = bbox[1]
    np.testing.assert_allclose(cloud[y1:y2, x1:x2].reshape(-1, 3)[::61], g["cloud_crop_smp"], rtol=1e-6, atol=1e-7)
    ch = np.arange(0, (y2 - y1) * (x2 - x1), 37)
    np.testing.assert_array_equal(opre.resize_rgb_choose(ch, [y1, y2, x1, x2], 224), g["rgb_choose"])
    # batched boxes of the product == the reference rule, mask by mask
    m = torch.from_numpy(np.logical_and(masks, depth > 0))
    np.testing.assert_array_equal(pre.square_boxes(m).numpy(), g["bbox"])


def test_batched_preprocessing_matches_oracle_loop():
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(radius=0.12, n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(),
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], keys=inp["keys"], **kw)
    assert out["kept"].tolist() == ref["kept"].tolist() and len(ref["kept"]) >= 6          # the 4x3-px proposal is dropped
    np.testing.assert_array_equal(out["bbox"].numpy(), ref["bbox"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb"].numpy(), ref["rgb"])
    # both sampler branches were exercised: a detection with fewer inliers than n_sample repeats points
    assert any(len(np.unique(r)) < 512 for r in ref["rgb_choose"]) and any(len(np.unique(c, axis=0)) == 512 for c in ref["pts"])


def test_no_detection_survives():
    inp = synth.pem_pre_inputs(P=3, seed=5)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"] * 0, inp["K"], inp["masks"], 0.1, inp["keys"])
    assert out["pts"].shape[0] == 0 and out["rgb"].shape == (0, 3, 224, 224) and out["kept"].numel() == 0


def test_numpy_rng_mode_reproduces_the_reference_draws():
    """rng= mode: the same np.random.choice calls, in the same order, as the reference's statements (golden made by
    executing run_inference_custom.py:224-229 and bop_test_dataset.py:140-145 with numpy's RNG seeded)."""
    g = util.golden("pem_pre.npz")
    np.testing.assert_array_equal(g["rng_idx_custom"], g["rng_idx_bop"])          # both entry points draw alike
    counts = torch.from_numpy(g["rng_counts"])
    ok = torch.tensor([True] * len(counts))
    np.random.seed(11)
    idx = pre._numpy_choice_indices(counts, ok, 512, np.random)
    np.testing.assert_array_equal(idx.numpy(), g["rng_idx_custom"])
    # a detection that failed a size test consumes no draw
    np.random.seed(11)
    ok2 = torch.tensor([True, False, True, True, True])
    idx2 = pre._numpy_choice_indices(counts, ok2, 512, np.random)
    np.testing.assert_array_equal(idx2[0].numpy(), g["rng_idx_custom"][0])
    assert (idx2[1] == 0).all() and not np.array_equal(idx2[2].numpy(), g["rng_idx_custom"][2])
    # whole frame: product (batched) == oracle loop with the same seeded stream
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(radius=0.12, n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(),
                                rng=np.random.RandomState(5), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], rng=np.random.RandomState(5), **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])
    import pytest
    with pytest.raises(ValueError):
        pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], 0.12)
    with pytest.raises(ValueError):
        pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], 0.12, keys=inp["keys"], rng=np.random)


def test_per_detection_radius_on_multi_object_frames():
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    # radii chosen so that no point lies within a few micrometres of a detection's radius sphere: the centroid is a sequential
    # float32 sum in the reference (np.mean over axis 0) and a float64-accumulated mean in the batched code, so points that
    # close to the boundary can fall on different sides (known gap)
    radius = np.array([0.12, 0.03, 0.5, 0.12, 0.05, 0.2, 0.12, 0.01])
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), radius,
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], torch.from_numpy(radius),
                              keys=inp["keys"], **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])
    same = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), 0.12,
                                 keys=inp["keys"].numpy(), **kw)
    assert not np.array_equal(same["pts"][1], ref["pts"][1])                  # the smaller radius really changed a cloud


def test_cv2_resize_restatement_properties():
    """oracle.pem_pre.cv2_resize_linear_u8 (the restated cv2.resize(INTER_LINEAR) of the PEM colour crop, f3): the properties
    OpenCV's algorithm has by construction -- 1:1 is a copy, 2:1 is the rounded 2x2 box mean, a constant image stays constant
    (the two independently rounded coefficients always add up to 2048 +- 1 and the >> 4 / >> 16 / + 2 >> 2 chain absorbs it),
    2x upsampling uses the 3/4 - 1/4 weights (1536 / 512 of 2048) away from the border, and the result is within one grey
    level of exact bilinear interpolation."""
    from oracle import pem_pre as o
    g = np.random.default_rng(3)
    img = g.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert np.array_equal(o.cv2_resize_linear_u8(img, 224), img)
    big = g.integers(0, 256, (448, 448, 3), dtype=np.uint8).astype(np.int32)
    want = (big[0::2, 0::2] + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2
    assert np.array_equal(o.cv2_resize_linear_u8(big.astype(np.uint8), 224), want.astype(np.uint8))
    for side in (37, 100, 333, 500):
        for v in (0, 1, 127, 254, 255):
            assert (o.cv2_resize_linear_u8(np.full((side, side, 3), v, np.uint8), 224) == v).all(), (side, v)
    small = g.integers(0, 256, (112, 112, 3), dtype=np.uint8)
    up = o.cv2_resize_linear_u8(small, 224).astype(np.int32)
    s = small.astype(np.int32)
    t = s[:, 5] * 1536 + s[:, 6] * 512                                 # output column 11 (centre 5.25) <- source columns 5, 6
    want = (((1536 * (t[5] >> 4)) >> 16) + ((512 * (t[6] >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(up[11, 11], want)
    for side in (37, 100, 333, 500, 96):
        im = g.integers(0, 256, (side, side, 3), dtype=np.uint8)
        c = np.clip((np.arange(224) + 0.5) * side / 224 - 0.5, 0, side - 1)
        i0 = np.floor(c).astype(int)
        i1, f = np.minimum(i0 + 1, side - 1), c - np.floor(c)
        rows = im[i0].astype(np.float64) * (1 - f)[:, None, None] + im[i1] * f[:, None, None]
        ref = rows[:, i0] * (1 - f)[None, :, None] + rows[:, i1] * f[None, :, None]
        assert np.abs(o.cv2_resize_linear_u8(im, 224) - ref).max() < 1.0


def test_cv2_vectors_if_present():
    """tests/golden/cv2_resize.npz (tools/gen_cv2_vectors.py, to be generated wherever cv2 is importable): the restatement
    against OpenCV's own outputs.  Absent in this image (no cv2 offline): the f3 colour crop stays 'parity unpinned'."""
    import hashlib
    import os

    import pytest
    path = os.path.join(util.GOLDEN, "cv2_resize.npz")
    if not os.path.exists(path):
        pytest.skip("no cv2 vectors yet (cv2 is not installable offline)")
    from oracle import pem_pre as o
    from tools.gen_cv2_vectors import image
    g = np.load(path)
    for s in g["sides"].tolist():
        out = o.cv2_resize_linear_u8(image(s, 1000 + s), 224)
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == str(g[f"sha_{s}"]), s


def test_rgb_choose_for_every_crop_size_in_float64_like_the_reference():
    """get_resize_rgb_choose (data_utils.py:113-123) multiplies by fl64(224 / crop): with crops of 140 or 160 pixels some
    row * ratio products land exactly on an integer, so a ratio one ulp off (torch's `scalar / tensor` = reciprocal * scalar)
    moves those rows by one -- every crop size a 480 x 640 frame can give, against the oracle helper (itself pinned to the
    reference's function by tests/golden/pem_pre.npz and example_frame.npz).  Found by the pixels-to-pose golden (round 5)."""
    for size in range(33, 481):
        # the map acts on the row and the column index separately: every row (column 0) and every column (row 0) of the crop
        ch = torch.cat([torch.arange(size, dtype=torch.int64) * size, torch.arange(1, size, dtype=torch.int64)])[None]
        box = torch.tensor([[0, size, 0, size]])
        out = pre._finish(torch.zeros(1, 1, 3, dtype=torch.uint8), None, box, torch.zeros(1, dtype=torch.int64), torch.zeros(1, 1, 3), ch, 224,
                          True, rgb=torch.zeros(1, 3, 224, 224))
        np.testing.assert_array_equal(out["rgb_choose"][0].numpy(), opre.resize_rgb_choose(ch[0].numpy(), [0, size, 0, size], 224), err_msg=str(size))
