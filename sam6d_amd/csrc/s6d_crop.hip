// Proposal crops for the DINOv2 descriptor model, fused (gfx950): normalise . mask . crop . nearest-resize . zero-pad .
// nearest-resize in one pass -- one read of the frame / mask pixels that survive, one write of the crop.
//
// Reference: CustomDINOv2.process_rgb_proposals / process_masks_proposals (Instance_Segmentation_Model/model/
// dinov2.py:131-144, 178-189) = rgb_normalize (ToTensor + Normalize), an (P,3,H,W) repeat * mask, then
// CropResizePad.__call__ (utils/bbox_utils.py:98-126): a per-proposal Python loop of slice, F.interpolate(nearest,
// scale_factor), F.pad, F.interpolate(nearest, scale_factor).  The host side (sam6d_amd/ism/dinov2.py) mirrors the
// loop's double-precision size arithmetic and hands every proposal's geometry over as one CropParams record; the
// kernel mirrors ATen's nearest index rule (UpSample.h nearest_neighbor_compute_source_index, the one the CUDA kernel
// and the CPU generic kernel use): src = min(floorf(dst * float(1 / scale)), in - 1) with the USER scale_factor, not
// out / in -- also when out == in (a 17-wide crop "resized" by 1.04 to 17 columns repeats column 0 and drops 16).
#include "s6d_common.h"

namespace s6d {

struct CropParams {     // 12 x 4 bytes; all sizes in pixels
  int x1, y1;           // crop origin in the frame
  int h, w;             // crop size (box[3]-box[1], box[2]-box[0]: the max corner is EXCLUDED, as in the reference)
  int h1, w1;           // size after the first resize: floor(h * s1), floor(w * s1)
  int top, left;        // zero padding in front of the resized crop
  int S2;               // side of the padded square
  float inv1, inv2;     // float(1 / s1), float(1 / s2): ATen's compute_scales_value<float>
  int pad_;
};

__device__ __forceinline__ int nearest_src(int dst, int in, float inv) {
  return min((int)floorf((float)dst * inv), in - 1);
}

__global__ __launch_bounds__(256) void crop_resize_pad_kernel(const unsigned char *__restrict__ image,
                                                              const float *__restrict__ masks,
                                                              const CropParams *__restrict__ params, int P, int H, int W,
                                                              int T, float m0, float m1, float m2, float s0, float s1,
                                                              float s2, float *__restrict__ out_rgb,
                                                              float *__restrict__ out_mask) {
  const int p = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= T * T) return;
  const CropParams c = params[p];
  const int oy = o / T, ox = o - oy * T;
  // second resize: padded square (S2) -> T
  const int py = nearest_src(oy, c.S2, c.inv2), px = nearest_src(ox, c.S2, c.inv2);
  // padding
  const int iy = py - c.top, ix = px - c.left;
  float r = 0.f, g = 0.f, b = 0.f, mk = 0.f;
  if (iy >= 0 && iy < c.h1 && ix >= 0 && ix < c.w1) {
    // first resize: crop (h, w) -> (h1, w1)
    const int sy = c.y1 + nearest_src(iy, c.h, c.inv1), sx = c.x1 + nearest_src(ix, c.w, c.inv1);
    const size_t pix = (size_t)sy * W + sx;
    mk = masks[(size_t)p * H * W + pix];
    if (out_rgb) {
      const unsigned char *q = image + pix * 3;
      // ToTensor (/255), Normalize ((x - mean) / std), then * mask: the reference's three roundings, in its order
      r = (((float)q[0] / 255.0f - m0) / s0) * mk;
      g = (((float)q[1] / 255.0f - m1) / s1) * mk;
      b = (((float)q[2] / 255.0f - m2) / s2) * mk;
    }
  }
  const size_t plane = (size_t)T * T;
  if (out_rgb) {
    float *d = out_rgb + (size_t)p * 3 * plane + o;
    d[0] = r;
    d[plane] = g;
    d[2 * plane] = b;
  }
  if (out_mask) out_mask[(size_t)p * plane + o] = mk;
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_crop_resize_pad_f32(const unsigned char *image, const float *masks, const void *params, int P, int H,
                                       int W, int T, const float *mean3_host, const float *std3_host, float *out_rgb,
                                       float *out_mask, void *stream) {
  if (P < 0 || H <= 0 || W <= 0 || T <= 0) return S6D_EINVAL;
  if (P == 0) return S6D_OK;
  if (!masks || !params || (!out_rgb && !out_mask) || (out_rgb && (!image || !mean3_host || !std3_host)))
    return S6D_EINVAL;
  static_assert(sizeof(CropParams) == 48, "CropParams is the 12-int record of include/sam6d_hip.h");
  const float one[3] = {1.f, 1.f, 1.f}, zero[3] = {0.f, 0.f, 0.f};
  const float *m = out_rgb ? mean3_host : zero, *s = out_rgb ? std3_host : one;
  const dim3 grid((unsigned)((T * T + 255) / 256), (unsigned)P);
  hipLaunchKernelGGL(crop_resize_pad_kernel, grid, dim3(256), 0, as_stream(stream), image, masks,
                     (const CropParams *)params, P, H, W, T, m[0], m[1], m[2], s[0], s[1], s[2], out_rgb, out_mask);
  return launch_status();
}
