"""Frame -> encoder-input resize of the SAM path (the caller side of the image encoder).

Reference: ``segment_anything/utils/transforms.py`` ResizeLongestSide (:16-102).  ``SamPredictor.set_image``
(predictor.py:56-59) calls ``apply_image``, i.e. ``np.array(torchvision.resize(to_pil_image(image), target_size))``: a PIL
BILINEAR resize of the uint8 frame -- Pillow's two-pass fixed-point resampler (libImaging/Resample.c: per output pixel a
window of ``ceil(support) * 2 + 1`` taps, triangle weights scaled by max(scale, 1) when shrinking, normalised in double,
quantised to 22 fractional bits, accumulated in int32 with a rounding half, clipped to uint8 after EACH pass, horizontal
pass first).  It is integer arithmetic once the coefficient tables exist, so it is restated exactly: the tables are built
on the host in double (a few thousand numbers per frame size, cached), the two passes run as integer tensor ops on the
device the frame lives on, and the result is bit-identical with Pillow (tests/test_host_sam_transforms.py checks it against
Pillow itself, which is installed here and on the GPU box, and against tests/golden/sam_transforms.npz).

``apply_image_torch`` (float bilinear with antialiasing) is kept for callers that used it; the reference notes that it
"may not exactly match apply_image. apply_image is the transformation expected by the model".
"""
from functools import lru_cache

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2                                        # Resample.c: 8 bits of data, 2 bits of headroom


@lru_cache(maxsize=64)
def pil_bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Resample.c for the triangle filter (support 1) over the whole axis.
    -> (first tap index per output pixel (out,) int64, quantised weights (out, ksize) int32; taps past the window are 0)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xmin = np.zeros(out_size, dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        lo = int(center - support + 0.5)                             # C (int) cast: truncation toward zero
        lo = max(lo, 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        x = np.arange(n, dtype=np.float64)
        w = np.abs((x + lo - center + 0.5) * ss)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = 0.0
        for v in w:                                                  # the C loop's summation order
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        kk[xx, :n] = w
        xmin[xx] = lo
    q = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)  # (int) cast
    return xmin, q.astype(np.int32)


def _resample_axis(img, axis_len_out, axis):
    """One pass of ImagingResampleHorizontal/Vertical_8bpc along `axis` (0 = rows, 1 = columns) of an (H,W,C) uint8
    tensor."""
    n_in = img.shape[axis]
    xmin, kk = pil_bilinear_coeffs(int(n_in), int(axis_len_out))
    dev = img.device
    xmin_t = torch.from_numpy(xmin).to(dev)
    kk_t = torch.from_numpy(kk).to(dev)
    ksize = kk_t.shape[1]
    acc = torch.full((axis_len_out,) + tuple(img.shape[:axis]) + tuple(img.shape[axis + 1:]), 1 << (PRECISION_BITS - 1),
                     dtype=torch.int32, device=dev)
    src = img.movedim(axis, 0).to(torch.int32)                       # (n_in, ...)
    shape_w = (axis_len_out,) + (1,) * (src.dim() - 1)
    for j in range(ksize):
        idx = (xmin_t + j).clamp(max=n_in - 1)                       # weights of taps beyond the window are zero
        acc += src[idx] * kk_t[:, j].reshape(shape_w)
    out = (acc >> PRECISION_BITS).clamp(0, 255).to(torch.uint8)      # clip8
    return out.movedim(0, axis)


def pil_bilinear_resize_u8(img, out_hw):
    """Pillow ``Image.resize((w, h), BILINEAR)`` of an (H,W,C) or (H,W) uint8 tensor, on its device, bit-exact."""
    if img.dtype != torch.uint8 or img.dim() not in (2, 3):
        raise ValueError("expected an (H,W,C) or (H,W) uint8 tensor")
    h, w = int(out_hw[0]), int(out_hw[1])
    if h <= 0 or w <= 0:
        raise ValueError("height and width must be > 0")             # Pillow's own message
    x = img if img.dim() == 3 else img[:, :, None]
    if (h, w) == tuple(x.shape[:2]):
        out = x.clone()                                              # Image.resize returns a copy for an identity resize
    else:
        out = x
        if w != x.shape[1]:
            out = _resample_axis(out, w, 1)                          # horizontal pass first, uint8 in between
        if h != x.shape[0]:
            out = _resample_axis(out, h, 0)
    return out if img.dim() == 3 else out[:, :, 0]


class ResizeLongestSide:
    """Drop-in for segment_anything.utils.transforms.ResizeLongestSide (same methods, same arithmetic).  ``apply_image``
    accepts the reference's numpy HxWxC uint8 array (returns numpy) or a uint8 tensor on any device (returns a tensor
    there: no host round trip in the frame loop)."""

    def __init__(self, target_length, device=None):
        self.target_length = target_length
        self.device = device

    def apply_image(self, image):
        target = self.get_preprocess_shape(image.shape[0], image.shape[1], self.target_length)
        if isinstance(image, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(image))
            if self.device is not None:
                t = t.to(self.device)
            return pil_bilinear_resize_u8(t, target).cpu().numpy()
        return pil_bilinear_resize_u8(image, target)

    def _factors(self, original_size):
        """(x factor, y factor) = resized extent / original extent, Python doubles as in the reference."""
        nh, nw = self.get_preprocess_shape(original_size[0], original_size[1], self.target_length)
        return nw / original_size[1], nh / original_size[0]

    def apply_coords(self, coords, original_size):
        """(..., 2) numpy (x, y) points of the original frame -> float64 points of the resized frame (:33-47)."""
        fx, fy = self._factors(original_size)
        out = np.array(coords, dtype=float, copy=True)
        out[..., 0] *= fx
        out[..., 1] *= fy
        return out

    def apply_boxes(self, boxes, original_size):
        """(B,4) numpy XYXY boxes (:49-55)."""
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    def apply_image_torch(self, image):
        """(B,C,H,W) float: antialiased bilinear interpolation (:57-68) -- NOT what set_image uses."""
        size = self.get_preprocess_shape(image.shape[2], image.shape[3], self.target_length)
        return torch.nn.functional.interpolate(image, size, mode="bilinear", align_corners=False, antialias=True)

    def apply_coords_torch(self, coords, original_size):
        """(..., 2) tensor -> float32 tensor (:70-85; the factors stay Python doubles, the product is float32)."""
        fx, fy = self._factors(original_size)
        out = coords.detach().clone().to(torch.float)
        out[..., 0] = out[..., 0] * fx
        out[..., 1] = out[..., 1] * fy
        return out

    def apply_boxes_torch(self, boxes, original_size):
        return self.apply_coords_torch(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    @staticmethod
    def get_preprocess_shape(oldh, oldw, long_side_length):
        scale = long_side_length * 1.0 / max(oldh, oldw)
        newh, neww = oldh * scale, oldw * scale
        return int(newh + 0.5), int(neww + 0.5)
