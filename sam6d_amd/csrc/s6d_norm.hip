// Residual-add + LayerNorm over the channel dim for the bf16 token maps of the SAM encoder (gfx950).
//
// Reference: Block.forward, segment_anything/modeling/image_encoder.py:166-182
//     x = shortcut + attn(...) ; x = x + mlp(norm2(x))       (and norm1 of the next block)
// Under reduced precision the reference (autocast) runs LayerNorm in fp32 with a cast copy on either
// side; here one pass reads x (and the branch output to add), writes the updated residual stream and
// the normalised activations, with fp32 statistics.  HBM-bound: 2 reads + 2 writes of C bf16 per token.
#include "s6d_common.h"

namespace s6d {

typedef unsigned short u16;
__device__ __forceinline__ float bf2f_(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ u16 f2bf_(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// one wavefront per token row; VEC chunks of 8 bf16 (16 B) per lane: C <= 64*8*VEC
template <int VEC>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const u16 *__restrict__ x, const u16 *__restrict__ delta,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, long rows,
                                                           int C, u16 *__restrict__ x_out, u16 *__restrict__ y_out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nchunk = C / 8;
  float v[VEC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = lane + i * 64;
    if (ch < nchunk) {
      union { uint4 u; u16 h[8]; } a, d;
      a.u = *reinterpret_cast<const uint4 *>(x + row * C + ch * 8);
      if (delta) {
        d.u = *reinterpret_cast<const uint4 *>(delta + row * C + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a.h[e] = f2bf_(bf2f_(a.h[e]) + bf2f_(d.h[e]));   // residual stream stays bf16
        if (x_out) *reinterpret_cast<uint4 *>(x_out + row * C + ch * 8) = a.u;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = bf2f_(a.h[e]);
        sum += v[i][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = lane + i * 64;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = lane + i * 64;
    if (ch < nchunk) {
      union { uint4 u; u16 h[8]; } o;
      const float4 g0 = *reinterpret_cast<const float4 *>(gamma + ch * 8), g1 = *reinterpret_cast<const float4 *>(gamma + ch * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4 *>(beta + ch * 8), b1 = *reinterpret_cast<const float4 *>(beta + ch * 8 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) o.h[e] = f2bf_((v[i][e] - mean) * rstd * gg[e] + bb[e]);
      *reinterpret_cast<uint4 *>(y_out + row * C + ch * 8) = o.u;
    }
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_add_layernorm_bf16(const void *x, const void *delta, const float *gamma, const float *beta,
                                      float eps, long rows, int C, void *x_out, void *y_out, void *stream) {
  if (rows < 0 || C <= 0 || (C % 8) != 0) return S6D_EINVAL;
  if (rows == 0) return S6D_OK;
  if (!x || !gamma || !beta || !y_out || (delta && !x_out)) return S6D_EINVAL;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int nchunk = C / 8;
  hipStream_t st = as_stream(stream);
#define S6D_LN(V)                                                                                              \
  hipLaunchKernelGGL((add_layernorm_kernel<V>), dim3(grid), dim3(256), 0, st, (const u16 *)x, (const u16 *)delta, \
                     gamma, beta, eps, rows, C, (u16 *)x_out, (u16 *)y_out)
  if (nchunk <= 64) S6D_LN(1);
  else if (nchunk <= 128) S6D_LN(2);
  else if (nchunk <= 192) S6D_LN(3);
  else if (nchunk <= 256) S6D_LN(4);
  else return S6D_EUNSUPPORTED;
#undef S6D_LN
  return launch_status();
}
