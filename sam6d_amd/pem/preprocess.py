"""Per-detection pre-processing of the Pose Estimation Model on the device.

Reference: ``Pose_Estimation_Model/run_inference_custom.py`` get_test_data :197-244 and ``provider/bop_test_dataset.py``
get_instance :113-156 -- for every detection of a frame: mask AND valid depth, square crop box (utils/data_utils.py
get_bbox :126-160), back-projection of the masked pixels (get_point_cloud_from_depth :92-110), radius filter around the
centroid, sampling of 2048 points, the masked colour crop resized to 224 x 224 and normalised, and the index of every
sampled point in that 224 x 224 grid (get_resize_rgb_choose :113-123).  The reference does this in 16 CPU DataLoader
workers, one Python loop iteration per detection; here all detections of a frame go through one set of batched tensor
ops on the device (ragged pixel lists are handled as one sorted (detection, y, x) list with per-detection offsets) and
two host round trips per FRAME (the pixel-list length and the survivor list), none per detection.

Defined differently from the reference, on purpose (oracle/pem_pre.py explains and mirrors both):
  * sampling uses INJECTED uniforms (``keys``, one per crop pixel) instead of numpy's global RNG: with replacement
    idx_i = floor(u_i * n) when n <= n_sample, else the n_sample smallest keys in key order.  (``rng=`` switches to the
    reference's own draws -- ``np.random.choice`` once per surviving detection, in detection order, :224-227 -- for runs that
    must reproduce a seeded reference run point for point; it costs one more device->host copy of P counts);
  * the colour crop: the reference calls cv2.resize(INTER_LINEAR) on the uint8 crop; cv2 is not in this image, so its published
    fixed-point algorithm (OpenCV 4.x resize.cpp: 11-bit coefficients, two passes, box average at 2:1, copy at 1:1) is
    restated integer for integer since round 3 -- still "parity unpinned" until tools/gen_cv2_vectors.py has been run where cv2
    exists (round 2 used a float32 bilinear that could differ by one grey level);
  * the radius test ``|p - centre| < radius * 1.2`` compares the float32 distance against the float64 product (what numpy does
    for a float64 ``radius``; for a Python-float radius numpy >= 2 (NEP 50, the version the goldens were made with: 2.2) rounds
    the product to float32 first, and numpy 1.x did so by value-based casting -- a point would have to lie within one float32
    ulp of the sphere to tell the three apart).
"""
import os

import torch

from .. import policy

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def square_boxes(m):
    """get_bbox for a stack of masks (P,H,W) bool with at least one set pixel each -> (P,4) long [rmin,rmax,cmin,cmax]."""
    P, H, W = m.shape
    rows, cols = m.any(2), m.any(1)
    ar_h, ar_w = torch.arange(H, device=m.device), torch.arange(W, device=m.device)
    rmin = torch.where(rows, ar_h, H).min(1).values
    rmax = torch.where(rows, ar_h, -1).max(1).values + 1
    cmin = torch.where(cols, ar_w, W).min(1).values
    cmax = torch.where(cols, ar_w, -1).max(1).values + 1
    b = torch.minimum(torch.maximum(rmax - rmin, cmax - cmin), torch.tensor(min(H, W), device=m.device))
    cy, cx, half = (rmin + rmax) // 2, (cmin + cmax) // 2, b // 2
    rmin, rmax, cmin, cmax = cy - half, cy + half, cx - half, cx + half
    d = (-rmin).clamp(min=0)
    rmin, rmax = rmin + d, rmax + d
    d = (-cmin).clamp(min=0)
    cmin, cmax = cmin + d, cmax + d
    d = (rmax - H).clamp(min=0)
    rmin, rmax = rmin - d, rmax - d
    d = (cmax - W).clamp(min=0)
    cmin, cmax = cmin - d, cmax - d
    return torch.stack([rmin, rmax, cmin, cmax], 1)


def _cv_taps(lo, hi, S, dev, clamp_index):
    """Per-axis tables of cv2.resize(INTER_LINEAR) on uint8 for P crops (OpenCV 4.x resize.cpp; see csrc/s6d_pempre.hip
    pem_crops_kernel and oracle/pem_pre.py cv2_resize_linear_u8): -> s (P,S) int64 source index, c0, c1 (P,S) int32."""
    n = (hi - lo).to(torch.float64)
    inv = torch.tensor(float(S), device=dev, dtype=torch.float64) / n                  # tensor divisors: IEEE quotients
    scale = torch.ones((), device=dev, dtype=torch.float64) / inv
    o = torch.arange(S, device=dev, dtype=torch.float64) + 0.5
    f = (o[None, :] * scale[:, None] - 0.5).to(torch.float32)
    s = f.floor()
    f = f - s
    s = s.long()
    if clamp_index:
        last = (hi - lo - 1)[:, None]
        lo_, hi_ = s < 0, s >= last
        f = torch.where(lo_ | hi_, torch.zeros_like(f), f)
        s = torch.where(lo_, torch.zeros_like(s), torch.where(hi_, last.expand_as(s), s))
    c0 = torch.round((1 - f) * 2048).clamp(-32768, 32767).to(torch.int32)            # torch.round = half to even = cvRound
    c1 = torch.round(f * 2048).clamp(-32768, 32767).to(torch.int32)
    return s, c0, c1


def _crops(image_u8, m, box, img_size, rgb_mask_flag):
    """Masked, channel-flipped colour crops resized with cv2.resize(INTER_LINEAR)'s fixed-point arithmetic and normalised:
    (P,3,S,S) f32 for boxes (P,4).  The library-op statement of pem_crops_kernel (integer for integer the same)."""
    P = box.shape[0]
    dev = image_u8.device
    S = img_size
    y1, y2, x1, x2 = box.unbind(1)
    h, w = y2 - y1, x2 - x1
    img = image_u8.flip(-1).to(torch.int32)                                           # [:, :, ::-1] of the reference
    mk = (m > 0).to(torch.int32)
    pidx = torch.arange(P, device=dev)[:, None, None]

    def px(yy, xx):                                                                    # (P,S,S,3) crop * mask, crop coordinates
        Y, X = (y1[:, None] + yy)[:, :, None], (x1[:, None] + xx)[:, None, :]
        v = img[Y, X]
        return v * mk[pidx, Y, X].unsqueeze(-1) if rgb_mask_flag else v
    sx, a0, a1 = _cv_taps(x1, x2, S, dev, True)
    sy, b0, b1 = _cv_taps(y1, y2, S, dev, False)
    ya, yb = sy.clamp(min=0).minimum((h - 1)[:, None]), (sy + 1).clamp(min=0).minimum((h - 1)[:, None])
    xb = (sx + 1).minimum((w - 1)[:, None])
    A0, A1 = a0[:, None, :, None], a1[:, None, :, None]
    t0 = px(ya, sx) * A0 + px(ya, xb) * A1
    t1 = px(yb, sx) * A0 + px(yb, xb) * A1
    lin = ((((b0[:, :, None, None] * (t0 >> 4)) >> 16) + ((b1[:, :, None, None] * (t1 >> 4)) >> 16) + 2) >> 2).clamp(0, 255)
    o = torch.arange(S, device=dev)
    ident = ((h == S) & (w == S))[:, None, None, None]
    half = ((h == 2 * S) & (w == 2 * S))[:, None, None, None]
    # the two special cases of cv2.resize: a copy at 1:1, the (a + b + c + d + 2) >> 2 box average at exactly 2:1 (indices are
    # clamped so that the unused branch of other crops stays in range)
    cp = px(o[None, :].minimum((h - 1)[:, None]), o[None, :].minimum((w - 1)[:, None]))
    e0y, e1y = (2 * o)[None, :].minimum((h - 1)[:, None]), (2 * o + 1)[None, :].minimum((h - 1)[:, None])
    e0x, e1x = (2 * o)[None, :].minimum((w - 1)[:, None]), (2 * o + 1)[None, :].minimum((w - 1)[:, None])
    area = (px(e0y, e0x) + px(e0y, e1x) + px(e1y, e0x) + px(e1y, e1x) + 2) >> 2
    out = torch.where(ident, cp, torch.where(half, area, lin)).to(torch.float32)
    c255 = torch.tensor(255.0, device=dev)
    mean = torch.tensor(MEAN, device=dev)
    std = torch.tensor(STD, device=dev)
    return ((out / c255 - mean) / std).permute(0, 3, 1, 2).contiguous()


@torch.no_grad()
def observed_inputs(image_u8, depth, K, masks, radius, keys=None, n_sample=2048, img_size=224, min_points=32, min_inliers=4,
                    radius_factor=1.2, rgb_mask_flag=True, rng=None):
    """image_u8 (H,W,3) uint8 RGB, depth (H,W) f32 metres, K 3x3 (host), masks (P,H,W) bool, radius: the object's radius in
    metres (a number, or a (P,) tensor with one radius per detection when a frame holds several objects), keys (P,H*W) f32 uniforms,
    all tensors on one device.  -> dict(pts (M,n,3) f32, rgb (M,3,S,S) f32, rgb_choose (M,n) i64, kept (M,) i64 indices
    of the detections that passed the two size tests (> min_points masked pixels, >= min_inliers after the radius
    filter), bbox (M,4) i64 [y1,y2,x1,x2]).  Exactly one of ``keys`` / ``rng`` (``numpy.random`` itself or a RandomState)
    selects the sampler."""
    if (keys is None) == (rng is None):
        raise ValueError("pass either keys (injected uniforms) or rng (numpy-compatible draws)")
    dev = depth.device
    P, H, W = masks.shape
    fx, fy, cx, cy = (float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
    if _use_kernels(depth):
        # kernel path (csrc/s6d_pempre.hip + s6d_segment_seq_sum_f32), the default on the device since its parity tests ran on an
        # MI355X (round 2): one pass per detection for mask AND depth / count / box, per-detection fixed-capacity lists instead of
        # one frame-wide list, the reference's sequential centroid, no host round trip before the survivor list.  A slot holds a
        # whole square crop (cap = min(H,W)^2 points, 16 B each): detections go through in chunks of at most _SLOT_BYTES of slots,
        # so a frame with hundreds of detections (BOP flow, top_k=None) or a 1080p frame stays bounded.
        cap = min(H, W) ** 2
        step = max(1, _SLOT_BYTES // (16 * cap))
        if P > step:
            outs = []
            for i in range(0, P, step):
                r = radius[i:i + step] if (torch.is_tensor(radius) and radius.numel() > 1) else radius
                o = observed_inputs(image_u8, depth, K, masks[i:i + step], r, None if keys is None else keys[i:i + step], n_sample,
                                    img_size, min_points, min_inliers, radius_factor, rgb_mask_flag, rng)
                o["kept"] = o["kept"] + i
                outs.append(o)
            return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
        from .. import ops
        mb = masks if masks.dtype == torch.bool else masks > 0
        m8, cnt, ok8, box = ops.pem_mask_boxes(mb.contiguous().view(torch.uint8), depth.contiguous(), min_points)
        m, ok1 = None, ok8.bool()
        choose_l, cloud_l, n = ops.pem_compact_cloud(m8, depth.contiguous(), box, ok8, fx, fy, cx, cy, cap)
        start = torch.arange(P, device=dev) * cap
        center = ops.segment_seq_sum(cloud_l.view(P * cap, 3), start, n) / n.clamp(min=1).float()[:, None]
        if torch.is_tensor(radius) and radius.numel() > 1:
            lim = (radius.to(dev).double() * radius_factor).contiguous()
        else:
            lim = torch.full((P,), float(radius) * radius_factor, dtype=torch.float64, device=dev)
        ops.pem_radius_filter(center.contiguous(), lim, choose_l, cloud_l, n)
        ok = ok1 & (n >= min_inliers)
        idx = _numpy_choice_indices(n, ok, n_sample, rng).to(dev) if rng is not None else _keyed_indices(n, keys, n_sample)
        kept = torch.nonzero(ok).squeeze(1)
        g = (start[:, None] + idx)[kept]
        rgb = ops.pem_crops(image_u8.contiguous(), m8, kept, box, img_size, rgb_mask_flag, MEAN, STD)
        return _finish(image_u8, m, box, kept, cloud_l.view(-1, 3)[g], choose_l.view(-1)[g].long(), img_size, rgb_mask_flag, rgb)
    # library-op path (S6D_PEM_PRE=library, and host tensors): the same quantities as batched tensor ops
    m = (masks > 0) & (depth > 0)[None]
    cnt = m.flatten(1).sum(1)
    ok1 = cnt > min_points
    box = square_boxes(m | ~ok1[:, None, None])                                        # dummy full mask where empty
    y1, y2, x1, x2 = box.unbind(1)
    # ---- ragged pixel lists as one (detection, y, x) list, row-major inside each crop ---------------------------------
    pyx = torch.nonzero(m)                                                             # host round trip #1 (list length)
    p_, y_, x_ = pyx.unbind(1)
    inside = (y_ >= y1[p_]) & (y_ < y2[p_]) & (x_ >= x1[p_]) & (x_ < x2[p_]) & ok1[p_]
    p_, y_, x_ = p_[inside], y_[inside], x_[inside]
    choose = (y_ - y1[p_]) * (x2 - x1)[p_] + (x_ - x1[p_])
    z = depth[y_, x_]
    fx_t, fy_t = torch.tensor(fx, device=dev), torch.tensor(fy, device=dev)            # tensor divisors: true division
    cloud = torch.stack([(x_.float() - cx) * z / fx_t, (y_.float() - cy) * z / fy_t, z], 1)  # float32, reference order
    n0 = torch.bincount(p_, minlength=P)
    # the reference's np.mean(cloud, axis=0): rows added in order into a float32 accumulator, then one float32 division
    center = _segment_seq_sum(cloud.contiguous(), (torch.cumsum(n0, 0) - n0).contiguous(), n0.contiguous()) \
        / n0.clamp(min=1).float()[:, None]
    dist = torch.linalg.norm(cloud - center[p_], dim=1)
    if torch.is_tensor(radius) and radius.numel() > 1:                                 # one radius per detection (multi-object frames)
        flag = dist.double() < (radius.to(dev).double() * radius_factor)[p_]
    else:
        flag = dist.double() < float(radius) * radius_factor
    p_, choose, cloud = p_[flag], choose[flag], cloud[flag]
    n = torch.bincount(p_, minlength=P)                                                # inliers per detection
    ok = ok1 & (n >= min_inliers)
    start = torch.cumsum(n, 0) - n                                                     # offset of each detection's list
    if rng is not None:
        idx = _numpy_choice_indices(n, ok, n_sample, rng).to(dev)
    else:
        idx = _keyed_indices(n, keys, n_sample)
    kept = torch.nonzero(ok).squeeze(1)
    g = (start[:, None] + idx)[kept].clamp(max=max(cloud.shape[0] - 1, 0))
    return _finish(image_u8, m, box, kept, cloud[g], choose[g], img_size, rgb_mask_flag)


def _finish(image_u8, m, box, kept, pts, ch, img_size, rgb_mask_flag, rgb=None):
    """Colour crops (unless the kernel path already made them) and the index of every sampled point in the resized crop, for
    the detections that survived."""
    dev = image_u8.device
    bk = box[kept]
    if rgb is None:
        rgb = _crops(image_u8, m[kept].float(), bk, img_size, rgb_mask_flag) if len(kept) else \
            torch.zeros(0, 3, img_size, img_size, device=dev)
    ch_h, ch_w = (bk[:, 1] - bk[:, 0]), (bk[:, 3] - bk[:, 2])
    row, col = ch // ch_w[:, None], ch % ch_w[:, None]
    # get_resize_rgb_choose (data_utils.py:113-123) in ITS float64 arithmetic: ratio = fl(img_size / crop) by a true division
    # (`scalar / tensor` is reciprocal() * scalar in torch: one ulp off for crops of 140 or 160 pixels, where row * ratio lands
    # on an integer -- found by the pixels-to-pose golden, round 5), then fl(row * ratio), floor
    size = torch.full((1,), float(img_size), dtype=torch.float64, device=dev)
    rgb_choose = ((row.double() * torch.div(size, ch_h.double())[:, None]).floor() * img_size +
                  (col.double() * torch.div(size, ch_w.double())[:, None]).floor()).long()
    return dict(pts=pts, rgb=rgb, rgb_choose=rgb_choose, kept=kept, bbox=bk)


def _keyed_indices(n, keys, n_sample):
    """The defined sampler: (P,n_sample) in-list positions from one uniform per crop pixel."""
    dev = n.device
    use_kernel = _use_kernels(n) and policy.current().pem_sampler != "library"
    if use_kernel and keys.dtype == torch.float32 and keys.is_contiguous() and n_sample <= 2048:
        # one workgroup per detection (s6d_pem_sample_indices_f32) instead of a top-k over a (P, L) table of 64-bit keys; no host
        # round trip for L
        from .. import ops
        idx, overflow = ops.pem_sample_indices(keys, n.contiguous(), n_sample)
        if not bool(overflow.any()):                                # heavily duplicated keys: the library path below
            return idx
    L = int(n.max().item()) if n.numel() else 0                                        # host round trip #2
    L = max(L, n_sample)
    kk = keys[:, :L].float()
    ar = torch.arange(L, device=dev)[None, :]
    # the n_sample smallest keys in (key, position) order == a stable argsort's first n_sample entries, by selection
    # instead of a full sort: non-negative float32 bit patterns order like the values, the position breaks ties
    comp = (kk.contiguous().view(torch.int32).long() << 32) | ar
    comp = torch.where(ar < n[:, None], comp, torch.full_like(comp, torch.iinfo(torch.int64).max))
    without = torch.topk(comp, n_sample, dim=1, largest=False, sorted=True).values & 0xFFFFFFFF
    with_r = (kk[:, :n_sample].double() * n[:, None]).floor().long()
    return torch.where((n <= n_sample)[:, None], with_r, without)


def _numpy_choice_indices(n, ok, n_sample, rng):
    """The reference's draws (run_inference_custom.py:224-227 / bop_test_dataset.py:140-145): for every detection that
    passed both size tests, in order, ONE ``choice`` over arange(n) -- with replacement when n <= n_sample, else without."""
    import numpy as np

    n_h, ok_h = n.cpu().tolist(), ok.cpu().tolist()
    idx = np.zeros((len(n_h), n_sample), dtype=np.int64)
    for i, (cnt, good) in enumerate(zip(n_h, ok_h)):
        if not good:
            continue
        if cnt <= n_sample:
            idx[i] = rng.choice(np.arange(cnt), n_sample)
        else:
            idx[i] = rng.choice(np.arange(cnt), n_sample, replace=False)
    return torch.from_numpy(idx)


_SLOT_BYTES = 1 << 30          # bound on the fixed-capacity point slots of one kernel-path call


def _use_kernels(t):
    """The csrc/s6d_pempre.hip path: device tensors, library present, not switched off (S6D_PEM_PRE=library)."""
    from .. import ops
    return t.is_cuda and policy.current().pem_pre != "library" and ops.have("pem_pre")


def _segment_seq_sum(x, start, count):
    """Row-order float32 segment sums: the device kernel, or -- for host tensors only (CPU tests of this module's tensor logic;
    the product path is the device) -- the same sequential float32 accumulation as a loop."""
    if x.is_cuda:
        from .. import ops
        return ops.segment_seq_sum(x, start, count)
    import numpy as np
    out = torch.zeros(start.shape[0], x.shape[1], dtype=torch.float32)
    xn = x.numpy()
    for i, (s0, c) in enumerate(zip(start.tolist(), count.tolist())):
        if c:                                                           # add.accumulate is strictly sequential: last row = the running sum
            out[i] = torch.from_numpy(np.add.accumulate(xn[s0:s0 + c], axis=0, dtype=np.float32)[-1].copy())
    return out

