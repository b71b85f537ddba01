"""CPU restatement of the PEM per-detection pre-processing -- TEST INFRASTRUCTURE ONLY.

numpy restatement of ``Pose_Estimation_Model/run_inference_custom.py`` get_test_data :197-244 (== provider/
bop_test_dataset.py get_instance :113-156) and of the helpers it calls in ``utils/data_utils.py``: get_bbox :126-160,
get_point_cloud_from_depth :92-110, get_resize_rgb_choose :113-123.  The three helpers are pinned by
tests/golden/pem_pre.npz (the reference functions run unmodified).  Two things cannot be pinned and say so:
  * the point sampling: the reference draws from numpy's global RNG (np.random.choice :224-227); here the random
    numbers are an INPUT (one uniform key per crop pixel): with replacement idx_i = floor(u_i * n), without replacement
    the n_sample smallest keys in key order -- the same distributions, a defined stream.  PARITY UNPINNED (RNG).
  * the 224 x 224 colour crop: the reference calls cv2.resize(INTER_LINEAR) on uint8 (cv2 is not in this image);
    restated here as bilinear interpolation with half-pixel centres in float32, rounded to uint8.  PARITY UNPINNED
    (cv2 fixed-point arithmetic may differ by one grey level).
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


def resize_bilinear_u8(img, size):
    """(h,w,3) uint8 -> (size,size,3) uint8, half-pixel centres, edge clamp, float32 arithmetic, round half up."""
    h, w = img.shape[:2]

    def taps(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
        i0 = np.floor(s).astype(np.int64)
        f = (s - i0).astype(np.float32)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f
    y0, y1, fy = taps(size, h)
    x0, x1, fx = taps(size, w)
    im = img.astype(np.float32)
    top = im[y0][:, x0] * (1 - fx)[None, :, None] + im[y0][:, x1] * fx[None, :, None]
    bot = im[y1][:, x0] * (1 - fx)[None, :, None] + im[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.floor(out + np.float32(0.5)), 0, 255).astype(np.uint8)


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
