// Small HBM-bound kernels at the edges of the hot path (gfx950).
//
//   sam_preprocess   Sam.preprocess, segment_anything/modeling/sam.py:164-174: (x - mean) / std, zero pad to the
//                    square encoder input -- fused with the cast to the encoder's compute dtype.
//   upsample_gather  ViT_AE.forward tail + get_chosen_pixel_feats, Pose_Estimation_Model/model/
//                    feature_extraction.py:111-114 + utils/model_utils.py:69-81: the reference pixel-shuffles the
//                    (B,196,16*256) up-projection to (B,256,56,56), bilinearly upsamples to 224x224 (51 MB per
//                    instance) and then reads 2048 pixels of it.  Here each chosen pixel interpolates its 4
//                    source texels straight out of the up-projection: 4 KB read + 1 KB written per point.
//   segment_seq_sum  numpy's reduction over axis 0 of a C-ordered (n, C >= 2) float32 array -- what the reference's
//                    ``np.mean(cloud, axis=0)`` computes for the centroid of a detection's point cloud
//                    (Pose_Estimation_Model/run_inference_custom.py:214, provider/bop_test_dataset.py:131): the rows are
//                    added one after the other into a float32 accumulator.  A parallel reduction rounds differently, and
//                    a centroid that is off by a micrometre moves points across the radius filter; so the order is kept:
//                    one wave per detection stages 512-row chunks in LDS (coalesced), lanes 0..C-1 add them in row order.
//   im2col3x3       the nine shifted views of a (B,H,W,C) two-byte map side by side, (B,H,W,9C) in (dy, dx, c) order with
//                    zeros outside the map: the A operand of the SAM neck's 3x3 convolution run as ONE GEMM over K = 9 C
//                    (segment_anything/modeling/image_encoder.py:91-97).  torch.cat over nine strided views wrote the same
//                    302 MB per 16 frames in 390 us; this pass is one coalesced 16-byte store per thread.
//   patchify         PatchEmbed's Conv2d with kernel = stride = p (image_encoder.py:375-395) is a GEMM over the pixels of a patch:
//                    (B,Cin,H,W) -> (B,H/p,W/p,Cin p p) in (c, dy, dx) order, the Conv2d weight's own flattening.
#include "s6d_common.h"
#include "s6d_seqsum.h"

namespace s6d {

typedef unsigned short u16;
__device__ __forceinline__ u16 f2bf_m(float f) {
  union { __bf16 b; u16 u; } x;
  x.b = (__bf16)f;
  return x.u;
}

// in (B,3,h,w) f32 -> out (B,3,S,S) bf16 or f32; pixels outside (h,w) are 0 (F.pad after normalisation)
template <bool BF16>
__global__ void sam_preprocess_kernel(const float *__restrict__ in, int B, int h, int w, int S, float m0, float m1,
                                      float m2, float is0, float is1, float is2, void *__restrict__ out) {
  const size_t total = (size_t)B * 3 * S * (S / 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x4 = (int)(i % (S / 4)) * 4;
    const size_t r = i / (S / 4);
    const int y = (int)(r % S);
    const int ch = (int)((r / S) % 3);
    const size_t b = r / ((size_t)3 * S);
    const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2), istd = ch == 0 ? is0 : (ch == 1 ? is1 : is2);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x4 + e;
      v[e] = (y < h && x < w) ? (in[((b * 3 + ch) * h + y) * (size_t)w + x] - mean) * istd : 0.f;
    }
    const size_t o = ((b * 3 + ch) * S + y) * (size_t)S + x4;
    if (BF16) {
      union { uint2 u; u16 hh[4]; } p;
#pragma unroll
      for (int e = 0; e < 4; ++e) p.hh[e] = f2bf_m(v[e]);
      *reinterpret_cast<uint2 *>(reinterpret_cast<u16 *>(out) + o) = p.u;
    } else {
      *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// up (B, G*G, P*P, C) f32: token (gy,gx), sub-pixel (sy,sx) -> pixel (gy*P+sy, gx*P+sx) of the (G*P)^2 map;
// choose (B,n) int64 pixel ids of the H x W image; out (B,n,C).  One wavefront per point, C/64 floats per lane.
__global__ __launch_bounds__(256) void upsample_gather_kernel(const float *__restrict__ up, const long *__restrict__ choose,
                                                             int B, int n, int G, int P, int C, int H, int W,
                                                             float *__restrict__ out) {
  const long pt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= (long)B * n) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(pt / n);
  const int S = G * P;
  const long pix = choose[pt];
  const int y = (int)(pix / W), x = (int)(pix - (long)y * W);
  // F.interpolate(mode="bilinear", align_corners=False): src = (dst + 0.5) * (S / size) - 0.5, clamped at 0
  float sy = ((float)y + 0.5f) * ((float)S / (float)H) - 0.5f, sx = ((float)x + 0.5f) * ((float)S / (float)W) - 0.5f;
  sy = fmaxf(sy, 0.f);
  sx = fmaxf(sx, 0.f);
  const int y0 = min((int)sy, S - 1), x0 = min((int)sx, S - 1);
  const int y1 = min(y0 + 1, S - 1), x1 = min(x0 + 1, S - 1);
  const float wy = sy - (float)y0, wx = sx - (float)x0;
  auto row = [&](int Y, int X) {
    const int tok = (Y / P) * G + (X / P), sub = (Y % P) * P + (X % P);
    return up + (((size_t)b * G * G + tok) * (P * P) + sub) * C;
  };
  const float *r00 = row(y0, x0), *r01 = row(y0, x1), *r10 = row(y1, x0), *r11 = row(y1, x1);
  for (int c0 = lane * 4; c0 < C; c0 += 256) {
    const float4 a = *reinterpret_cast<const float4 *>(r00 + c0), bq = *reinterpret_cast<const float4 *>(r01 + c0);
    const float4 cq = *reinterpret_cast<const float4 *>(r10 + c0), d = *reinterpret_cast<const float4 *>(r11 + c0);
    float4 o;
    // same association as the library statement: top = a(1-wx) + b wx; bot likewise; out = top(1-wy) + bot wy
    o.x = (a.x * (1.f - wx) + bq.x * wx) * (1.f - wy) + (cq.x * (1.f - wx) + d.x * wx) * wy;
    o.y = (a.y * (1.f - wx) + bq.y * wx) * (1.f - wy) + (cq.y * (1.f - wx) + d.y * wx) * wy;
    o.z = (a.z * (1.f - wx) + bq.z * wx) * (1.f - wy) + (cq.z * (1.f - wx) + d.z * wx) * wy;
    o.w = (a.w * (1.f - wx) + bq.w * wx) * (1.f - wy) + (cq.w * (1.f - wx) + d.w * wx) * wy;
    *reinterpret_cast<float4 *>(out + pt * C + c0) = o;
  }
}

// in (B,H,W,C) -> out (B,H,W,9C); C8 = C / 8 (one thread = 16 bytes of the output)
__global__ __launch_bounds__(256) void im2col3x3_kernel(const uint4 *__restrict__ in, int B, int H, int W, int C8,
                                                       uint4 *__restrict__ out) {
  const size_t total = (size_t)B * H * W * 9 * C8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int tap = (int)(r % 9);
    r /= 9;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const size_t b = r / H;
    const int sy = y + tap / 3 - 1, sx = x + tap % 3 - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = in[((b * H + sy) * (size_t)W + sx) * C8 + c8];
    out[i] = v;
  }
}

// in (B,Cin,H,W) -> out (B,H/p,W/p,Cin*p*p), element (c, dy, dx) of a patch at c*p*p + dy*p + dx; p8 = p / 8
__global__ __launch_bounds__(256) void patchify_kernel(const uint4 *__restrict__ in, int B, int Cin, int H, int W, int p,
                                                      uint4 *__restrict__ out) {
  const int p8 = p / 8, gy = H / p, gx = W / p, per = Cin * p * p8;          // 16-byte chunks per patch
  const size_t total = (size_t)B * gy * gx * per;
  const size_t W8 = (size_t)W / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % per);
    size_t r = i / per;
    const int px = (int)(r % gx);
    r /= gx;
    const int py = (int)(r % gy);
    const size_t b = r / gy;
    const int d8 = e % p8, dy = (e / p8) % p, c = e / (p8 * p);
    out[i] = in[((b * Cin + c) * H + (size_t)py * p + dy) * W8 + (size_t)px * p8 + d8];
  }
}

__global__ void nonfinite_clear_kernel(int *flags, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) flags[i] = 0;
}

// flags[b] = 1 if row b of x (B rows of n floats, n % 4 == 0) holds an inf or a NaN (exponent field all ones); flags zeroed by the launcher
__global__ __launch_bounds__(256) void nonfinite_rows_kernel(const float4 *__restrict__ x, long n4, int *__restrict__ flags) {
  const int b = blockIdx.y;
  const uint4 *row = reinterpret_cast<const uint4 *>(x) + (size_t)b * n4;
  unsigned bad = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint4 v = row[i];
    bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
           ((v.w & 0x7f800000u) == 0x7f800000u);
  }
  if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flags + b, 1);
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_nonfinite_rows_f32(const float *x, int B, long n, int32_t *flags, void *stream) {
  if (B < 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (n <= 0 || (n % 4) != 0 || !x || !flags || ((uintptr_t)x & 15)) return S6D_EINVAL;
  hipStream_t st = as_stream(stream);
  // the flags are cleared by a kernel, not hipMemsetAsync: the PEM forward is captured into a hipGraph by the frame pipeline, and a
  // memset node inside that capture left the flags unset on replay (every instance read as overflowed: measured, round 4)
  hipLaunchKernelGGL(nonfinite_clear_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, flags, B);
  const long n4 = n / 4;
  long g = (n4 + 256 * 8 - 1) / (256 * 8);
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(nonfinite_rows_kernel, dim3((unsigned)g, (unsigned)B), dim3(256), 0, st, (const float4 *)x, n4, flags);
  return launch_status();
}

extern "C" int s6d_im2col3x3_b16(const void *in, int B, int H, int W, int C, void *out, void *stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!in || !out) return S6D_EINVAL;
  const size_t total = (size_t)B * H * W * 9 * (C / 8);
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (const uint4 *)in, B, H, W, C / 8,
                     (uint4 *)out);
  return launch_status();
}

extern "C" int s6d_patchify_b16(const void *in, int B, int Cin, int H, int W, int p, void *out, void *stream) {
  if (B < 0 || Cin <= 0 || H <= 0 || W <= 0 || p <= 0 || (p % 8) != 0 || (H % p) != 0 || (W % p) != 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!in || !out) return S6D_EINVAL;
  const size_t total = (size_t)B * Cin * H * (W / 8);
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (const uint4 *)in, B, Cin, H, W, p,
                     (uint4 *)out);
  return launch_status();
}

extern "C" int s6d_sam_preprocess_f32(const float *in, int B, int h, int w, int S, const float *mean3_host,
                                      const float *std3_host, int out_bf16, void *out, void *stream) {
  if (B < 0 || h <= 0 || w <= 0 || S <= 0 || h > S || w > S || (S % 4) != 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!in || !out || !mean3_host || !std3_host) return S6D_EINVAL;
  const size_t total = (size_t)B * 3 * S * (S / 4);
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipStream_t st = as_stream(stream);
  if (out_bf16)
    hipLaunchKernelGGL((sam_preprocess_kernel<true>), dim3((unsigned)g), dim3(256), 0, st, in, B, h, w, S, mean3_host[0],
                       mean3_host[1], mean3_host[2], 1.f / std3_host[0], 1.f / std3_host[1], 1.f / std3_host[2], out);
  else
    hipLaunchKernelGGL((sam_preprocess_kernel<false>), dim3((unsigned)g), dim3(256), 0, st, in, B, h, w, S, mean3_host[0],
                       mean3_host[1], mean3_host[2], 1.f / std3_host[0], 1.f / std3_host[1], 1.f / std3_host[2], out);
  return launch_status();
}

extern "C" int s6d_upsample_gather_f32(const float *up, const int64_t *choose, int B, int n, int G, int P, int C, int H,
                                       int W, float *out, void *stream) {
  if (B < 0 || n < 0 || G <= 0 || P <= 0 || C <= 0 || (C % 4) != 0 || H <= 0 || W <= 0) return S6D_EINVAL;
  if ((size_t)B * n == 0) return S6D_OK;
  if (!up || !choose || !out) return S6D_EINVAL;
  const long pts = (long)B * n;
  hipLaunchKernelGGL(upsample_gather_kernel, dim3((unsigned)((pts + 3) / 4)), dim3(256), 0, as_stream(stream), up,
                     (const long *)choose, B, n, G, P, C, H, W, out);
  return launch_status();
}

extern "C" int s6d_segment_seq_sum_f32(const float *x, const int64_t *start, const int64_t *count, int P, int C, float *out,
                                       void *stream) {
  if (P < 0 || C < 2 || C > 4) return S6D_EINVAL;  // C = 1: numpy reduces a contiguous axis pairwise, not in row order
  if (P == 0) return S6D_OK;
  if (!x || !start || !count || !out) return S6D_EINVAL;
  hipLaunchKernelGGL(segment_seq_sum_kernel, dim3((unsigned)P), dim3(64), 0, as_stream(stream), x, (const long *)start,
                     (const long *)count, C, out);
  return launch_status();
}
