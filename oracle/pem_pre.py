"""CPU restatement of the PEM per-detection pre-processing -- TEST INFRASTRUCTURE ONLY.

numpy restatement of ``Pose_Estimation_Model/run_inference_custom.py`` get_test_data :197-244 (== provider/
bop_test_dataset.py get_instance :113-156) and of the helpers it calls in ``utils/data_utils.py``: get_bbox :126-160,
get_point_cloud_from_depth :92-110, get_resize_rgb_choose :113-123.  The three helpers are pinned by
tests/golden/pem_pre.npz (the reference functions run unmodified).  Two things cannot be pinned and say so:
  * the point sampling: the reference draws from numpy's global RNG (np.random.choice :224-227); here the random
    numbers are an INPUT (one uniform key per crop pixel): with replacement idx_i = floor(u_i * n), without replacement
    the n_sample smallest keys in key order -- the same distributions, a defined stream.  PARITY UNPINNED (RNG).
  * the 224 x 224 colour crop: the reference calls cv2.resize(INTER_LINEAR) on uint8.  cv2 is an un-vendored dependency and
    not in this image; since round 3 its PUBLISHED algorithm is restated exactly (cv2_resize_linear_u8: OpenCV 4.x
    modules/imgproc/src/resize.cpp -- 11-bit fixed-point coefficients, two passes, the 2x2 box average OpenCV substitutes at
    an exact 2:1 ratio, a plain copy at 1:1).  PARITY UNPINNED until vectors exist: tools/gen_cv2_vectors.py writes them
    wherever cv2 can be imported and tests/test_host_pem_pre.py consumes tests/golden/cv2_resize.npz when present.
Arithmetic types follow the reference under the NumPy 1.x it was released for (float32 arrays stay float32 when
combined with the float64 camera scalars).
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def get_bbox(label):
    """data_utils.get_bbox: square box of side min(max(extent), min(H, W)) around the mask, shifted into the image."""
    H, W = label.shape
    rows, cols = np.any(label, axis=1), np.any(label, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    rmax, cmax = rmax + 1, cmax + 1
    b = min(max(rmax - rmin, cmax - cmin), min(H, W))
    cy, cx = int((rmin + rmax) / 2), int((cmin + cmax) / 2)
    rmin, rmax, cmin, cmax = cy - int(b / 2), cy + int(b / 2), cx - int(b / 2), cx + int(b / 2)
    if rmin < 0:
        rmin, rmax = 0, rmax - rmin
    if cmin < 0:
        cmin, cmax = 0, cmax - cmin
    if rmax > H:
        rmin, rmax = rmin - (rmax - H), H
    if cmax > W:
        cmin, cmax = cmin - (cmax - W), W
    return [rmin, rmax, cmin, cmax]


def point_cloud(depth, K):
    """get_point_cloud_from_depth without bbox: (H,W,3) float32."""
    fx, fy, cx, cy = (np.float32(K[0, 0]), np.float32(K[1, 1]), np.float32(K[0, 2]), np.float32(K[1, 2]))
    H, W = depth.shape
    xmap, ymap = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    z = depth.astype(np.float32)
    return np.stack([(xmap - cx) * z / fx, (ymap - cy) * z / fy, z], axis=-1)


def resize_rgb_choose(choose, bbox, img_size):
    rmin, rmax, cmin, cmax = bbox
    ratio_h, ratio_w = img_size / (rmax - rmin), img_size / (cmax - cmin)
    row, col = choose // (cmax - cmin), choose % (cmax - cmin)
    return (np.floor(row * ratio_h) * img_size + np.floor(col * ratio_w)).astype(np.int64)


def sample_indices(n, n_sample, keys):
    """The defined sampler (see the module docstring): keys = this detection's uniforms, one per crop pixel."""
    if n <= n_sample:
        return np.floor(keys[:n_sample].astype(np.float64) * n).astype(np.int64)
    return np.argsort(keys[:n], kind="stable")[:n_sample]


def _cv_round_to_short(v):
    """saturate_cast<short>(float): cvRound = round half to even (SSE cvtss2si / lrintf), then saturation."""
    return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int32)


def cv2_linear_tables(n_out, n_in):
    """The per-axis tables resize() builds for INTER_LINEAR on CV_8U (resize.cpp, `resize_` in namespace cv::hal):
        scale = 1. / (double(n_out) / n_in);  f = (float)((d + 0.5) * scale - 0.5);  s = cvFloor(f);  f -= s
    x axis only (:~3990-4010): s < 0 -> f = 0, s = 0;  s >= n_in - 1 -> f = 0, s = n_in - 1.  The y axis keeps (s, f) and
    clips the two ROW INDICES instead (resizeGeneric_Invoker: clip(sy + k, 0, n_in)).  Coefficients: saturate_cast<short>
    ((1.f - f) * 2048), saturate_cast<short>(f * 2048) -- rounded independently (they need not add up to 2048).
    -> (s (n_out,) int64, c0, c1 (n_out,) int32) unclamped, as the y axis uses them; `clamp_x` applies the x-axis rule."""
    inv = np.float64(n_out) / np.float64(n_in)
    scale = np.float64(1.0) / inv
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coefs(f):
    return _cv_round_to_short((np.float32(1.0) - f) * np.float32(2048)), _cv_round_to_short(f * np.float32(2048))


def cv2_resize_linear_u8(img, size):
    """cv2.resize(img, (size, size), interpolation=cv2.INTER_LINEAR) for an (h, w, C) uint8 image, restated from OpenCV 4.x
    resize.cpp (IPP is bypassed for 8-bit linear unless useIPP_NotExact is set; there is no other special path on x86):
      * h == w == size: copy;
      * scale exactly 2 on both axes: INTER_LINEAR is replaced by the fast area average (src[2y][2x] + src[2y][2x+1] +
        src[2y+1][2x] + src[2y+1][2x+1] + 2) >> 2  (ResizeAreaFastVec<uchar>);
      * otherwise HResizeLinear<uchar,int,short,2048>: t = S[sx] * a0 + S[sx + 1] * a1 (int32) per source row, then
        VResizeLinear<uchar,...>:  dst = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2."""
    h, w = img.shape[:2]
    if h == size and w == size:
        return img.copy()
    inv_x, inv_y = np.float64(size) / w, np.float64(size) / h
    sx_, sy_ = 1.0 / inv_x, 1.0 / inv_y
    if abs(sx_ - 2) < np.finfo(np.float64).eps and abs(sy_ - 2) < np.finfo(np.float64).eps and int(sx_) == 2 and int(sy_) == 2:
        s = img.astype(np.int32)
        return ((s[0:2 * size:2, 0:2 * size:2] + s[0:2 * size:2, 1:2 * size:2] + s[1:2 * size:2, 0:2 * size:2]
                 + s[1:2 * size:2, 1:2 * size:2] + 2) >> 2).astype(np.uint8)
    xs, fx = cv2_linear_tables(size, w)
    lo, hi = xs < 0, xs >= w - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    xs = np.where(lo, 0, np.where(hi, w - 1, xs))
    a0, a1 = _coefs(fx)
    ys, fy = cv2_linear_tables(size, h)
    b0, b1 = _coefs(fy)
    y0, y1 = np.clip(ys, 0, h - 1), np.clip(ys + 1, 0, h - 1)
    S = img.astype(np.int32)
    x1 = np.minimum(xs + 1, w - 1)                                   # a1 == 0 wherever xs + 1 would leave the row
    t = S[:, xs] * a0[None, :, None] + S[:, x1] * a1[None, :, None]      # (h, size, C) int32: the horizontal pass
    out = (((b0[:, None, None] * (t[y0] >> 4)) >> 16) + ((b1[:, None, None] * (t[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, size):
    """The colour-crop resize of the PEM input = cv2.resize(INTER_LINEAR) (run_inference_custom.py:234)."""
    return cv2_resize_linear_u8(img, size)


def preprocess_frame(image_u8, depth, K, masks, radius, keys=None, n_sample=2048, img_size=224, min_points=32, min_inliers=4,
                     radius_factor=1.2, rgb_mask_flag=True, rng=None):
    """get_test_data's per-detection loop.  image_u8 (H,W,3) RGB, depth (H,W) f32 metres, masks (P,H,W) bool, keys
    (P,H*W) uniforms (or rng = numpy.random / a RandomState for the reference's np.random.choice draws).  -> dict(pts (M,n,3) f32, rgb (M,3,S,S) f32 normalised BGR->RGB flipped like the reference,
    rgb_choose (M,n) i64, kept (M,) indices of the detections that survived the two size tests)."""
    whole = point_cloud(depth, K)
    out = dict(pts=[], rgb=[], rgb_choose=[], kept=[], bbox=[])
    for p in range(masks.shape[0]):
        mask = np.logical_and(masks[p] > 0, depth > 0)
        if np.sum(mask) <= min_points:
            continue
        y1, y2, x1, x2 = get_bbox(mask)
        m = mask[y1:y2, x1:x2]
        choose = m.astype(np.float32).flatten().nonzero()[0]
        cloud = whole[y1:y2, x1:x2, :].reshape(-1, 3)[choose, :]
        center = np.mean(cloud, axis=0)
        r_p = radius[p] if np.ndim(radius) else radius                          # per-detection radius on multi-object frames
        flag = np.linalg.norm(cloud - center[None, :], axis=1) < r_p * radius_factor
        if np.sum(flag) < min_inliers:
            continue
        choose, cloud = choose[flag], cloud[flag]
        if rng is not None:                                                     # the reference's own draws, :224-227
            if len(choose) <= n_sample:
                idx = rng.choice(np.arange(len(choose)), n_sample)
            else:
                idx = rng.choice(np.arange(len(choose)), n_sample, replace=False)
        else:
            idx = sample_indices(len(choose), n_sample, keys[p])
        choose, cloud = choose[idx], cloud[idx]
        rgb = image_u8[y1:y2, x1:x2, :][:, :, ::-1]
        if rgb_mask_flag:
            rgb = rgb * (m[:, :, None] > 0).astype(np.uint8)
        rgb = resize_bilinear_u8(rgb, img_size)
        t = (rgb.astype(np.float32) / np.float32(255) - MEAN) / STD            # ToTensor + Normalize
        out["pts"].append(cloud.astype(np.float32))
        out["rgb"].append(t.transpose(2, 0, 1))
        out["rgb_choose"].append(resize_rgb_choose(choose, [y1, y2, x1, x2], img_size))
        out["kept"].append(p)
        out["bbox"].append([y1, y2, x1, x2])
    return {k: (np.stack(v) if len(v) else np.zeros((0,))) for k, v in out.items()}
