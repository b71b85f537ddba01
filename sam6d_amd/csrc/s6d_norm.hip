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

// element conversions of the 2-byte rows: bf16 (SAM / DINOv2 / the throughput option of the PEM ViT-B) or IEEE half (PEM ViT-B)
template <bool F16>
__device__ __forceinline__ float elem2f(u16 h) {
  if (F16) {
    union { u16 u; _Float16 f; } c;
    c.u = h;
    return (float)c.f;
  }
  return bf2f_(h);
}
template <bool F16>
__device__ __forceinline__ u16 f2elem(float f) {
  if (F16) {
    union { u16 u; _Float16 f; } c;
    c.f = (_Float16)f;
    return c.u;
  }
  return f2bf_(f);
}

// one wavefront per token row; VEC chunks of 8 bf16 (16 B) per lane: C <= 64*8*VEC
template <int VEC, bool F16>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const u16 *__restrict__ x, const u16 *__restrict__ delta,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, long rows,
                                                           int C, u16 *__restrict__ x_out, u16 *__restrict__ y_out,
                                                           float *__restrict__ y_f32) {
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
        for (int e = 0; e < 8; ++e) a.h[e] = f2elem<F16>(elem2f<F16>(a.h[e]) + elem2f<F16>(d.h[e]));   // residual stream stays bf16
        if (x_out) *reinterpret_cast<uint4 *>(x_out + row * C + ch * 8) = a.u;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = elem2f<F16>(a.h[e]);
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
      if (y_f32) {                                     // fp32 result (the SAM neck's last LayerNorm2d: the encoder's output)
        float4 o0, o1;
        o0.x = (v[i][0] - mean) * rstd * gg[0] + bb[0]; o0.y = (v[i][1] - mean) * rstd * gg[1] + bb[1];
        o0.z = (v[i][2] - mean) * rstd * gg[2] + bb[2]; o0.w = (v[i][3] - mean) * rstd * gg[3] + bb[3];
        o1.x = (v[i][4] - mean) * rstd * gg[4] + bb[4]; o1.y = (v[i][5] - mean) * rstd * gg[5] + bb[5];
        o1.z = (v[i][6] - mean) * rstd * gg[6] + bb[6]; o1.w = (v[i][7] - mean) * rstd * gg[7] + bb[7];
        *reinterpret_cast<float4 *>(y_f32 + row * C + ch * 8) = o0;
        *reinterpret_cast<float4 *>(y_f32 + row * C + ch * 8 + 4) = o1;
        continue;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o.h[e] = f2elem<F16>((v[i][e] - mean) * rstd * gg[e] + bb[e]);
      *reinterpret_cast<uint4 *>(y_out + row * C + ch * 8) = o.u;
    }
  }
}


// LayerNorm with an fp8 (OCP e4m3) output row and one power-of-two scale per row -- the activation operand of the fp8 GEMMs
// (csrc/s6d_gemm.hip gemm_fp8_kernel; BASELINE configs[4]).  y = LN(x) in fp32, amax over the row, scale 2^e with
// e = the smallest integer for which amax / 2^e <= 448 (e4m3's largest finite value), q = e4m3(y / 2^e) (round to nearest even),
// scale byte = e + 127 (E8M0, the MX block-scale encoding the matrix instruction takes).  An all-zero row gets byte 127.
// One wavefront per row, as add_layernorm_kernel.
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const u16 *__restrict__ x, const u16 *__restrict__ delta,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, long rows, int C,
                                                           u16 *__restrict__ x_out, unsigned char *__restrict__ y8,
                                                           unsigned char *__restrict__ yscale) {
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
      if (delta) {                                      // residual add folded in, as add_layernorm_kernel
        d.u = *reinterpret_cast<const uint4 *>(delta + row * C + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a.h[e] = f2bf_(bf2f_(a.h[e]) + bf2f_(d.h[e]));
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
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = lane + i * 64;
    if (ch < nchunk) {
      const float4 g0 = *reinterpret_cast<const float4 *>(gamma + ch * 8), g1 = *reinterpret_cast<const float4 *>(gamma + ch * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4 *>(beta + ch * 8), b1 = *reinterpret_cast<const float4 *>(beta + ch * 8 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
        amax = fmaxf(amax, fabsf(v[i][e]));
      }
    }
  }
  amax = wave_max(amax);
  // amax = f 2^ex with f in [0.5, 1): amax / 2^(ex - 9) = 512 f <= 448 iff f <= 0.875
  int e2 = 0;
  if (amax > 0.f) {
    int ex;
    const float f = frexpf(amax, &ex);
    e2 = ex - (f <= 0.875f ? 9 : 8);
    e2 = min(max(e2, -127), 127);
  }
  const float inv = ldexpf(1.f, -e2);
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = lane + i * 64;
    if (ch < nchunk) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, hi, true);
      *reinterpret_cast<uint2 *>(y8 + row * C + ch * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    }
  }
  if (lane == 0) yscale[row] = (unsigned char)(e2 + 127);
}


// ---- row statistics of the FOLDED LayerNorm (csrc/s6d_gemm.hip: EPI 2 writes partials, EPI 3 / 4 consume (mean, sigma)) ---------
// sigma = sqrt(var + eps): the consuming GEMM starts its accumulators at sigma b' - mean s and multiplies by 1 / sigma at the end.
// Partials: for every row and group of `gsz` columns the sum and the sum of squared deviations from the group mean, laid out
// [group][2][M].  Combined group by group in index order with the exact pairwise update (Chan et al.): no cancellation, fixed order.
// Latency, not bandwidth, is what this pass costs (20 MB): a thread's 2 x groups loads are issued eight groups at a time ahead of the
// (sequential) combination, 64-thread workgroups spread the rows over all CUs.
// Round 5: ALL partials of a row are requested before the first combination (the 40-group case of the ViT-H stream in one template
// instantiation: 80 registers); in groups of eight the pass was five dependent L2 round trips = 12.4 us per launch, 126 launches per step.
template <int GROUPS>
__global__ __launch_bounds__(64) void ln_stats_finalize_fixed_kernel(const float *__restrict__ sp, int gsz, long M, float eps,
                                                                    float *__restrict__ out) {
  const long row = (long)blockIdx.x * 64 + threadIdx.x;
  if (row >= M) return;
  float s[GROUPS], q[GROUPS];
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    s[g] = sp[(size_t)(2 * g) * M + row];
    q[g] = sp[(size_t)(2 * g + 1) * M + row];
  }
  const float nb = (float)gsz;
  float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {                    // the same pairwise update in the same order as the generic kernel below
    const float d = s[g] / nb - mean, nn = n + nb;
    mean += d * (nb / nn);
    m2 += q[g] + d * d * (n * nb / nn);
    n = nn;
  }
  *reinterpret_cast<float2 *>(out + row * 2) = make_float2(mean, sqrtf(m2 / n + eps));
}

__global__ __launch_bounds__(64) void ln_stats_finalize_kernel(const float *__restrict__ sp, int groups, int gsz, long M, float eps,
                                                              float *__restrict__ out) {
  const long row = (long)blockIdx.x * 64 + threadIdx.x;
  if (row >= M) return;
  const float nb = (float)gsz;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int g0 = 0; g0 < groups; g0 += 8) {
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int g = min(g0 + i, groups - 1);
      s[i] = sp[(size_t)(2 * g) * M + row];
      q[i] = sp[(size_t)(2 * g + 1) * M + row];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (g0 + i < groups) {
        const float d = s[i] / nb - mean, nn = n + nb;
        mean += d * (nb / nn);
        m2 += q[i] + d * d * (n * nb / nn);
        n = nn;
      }
    }
  }
  *reinterpret_cast<float2 *>(out + row * 2) = make_float2(mean, sqrtf(m2 / n + eps));
}

// the same (mean, sigma) straight from a bf16 matrix: one wavefront per row, two passes over registers (the statistics of
// add_layernorm_kernel).  Used once per encoder pass, for the patch-embedding output that enters block 0.
template <int VEC>
__global__ __launch_bounds__(256) void row_stats_kernel(const u16 *__restrict__ x, long ldx, long rows, int C, float eps,
                                                       float *__restrict__ out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nchunk = C / 8;
  float v[VEC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = lane + i * 64;
    union { uint4 u; u16 h[8]; } a;
    a.u = make_uint4(0u, 0u, 0u, 0u);
    if (ch < nchunk) a.u = *reinterpret_cast<const uint4 *>(x + row * ldx + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[i][e] = bf2f_(a.h[e]);
      sum += v[i][e];
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
  const float sigma = sqrtf(wave_sum(sq) / (float)C + eps);
  if (lane == 0) *reinterpret_cast<float2 *>(out + row * 2) = make_float2(mean, sigma);
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_ln_stats_finalize(const float *stats_partial, int groups, int group_size, long M, float eps, float *row_stats,
                                     void *stream) {
  if (M < 0 || groups <= 0 || group_size <= 0) return S6D_EINVAL;
  if (M == 0) return S6D_OK;
  if (!stats_partial || !row_stats || ((uintptr_t)row_stats & 7)) return S6D_EINVAL;
  const dim3 grid((unsigned)((M + 63) / 64));
  if (groups == 40)                      // ViT-H (1280 channels): every partial in flight at once
    hipLaunchKernelGGL(ln_stats_finalize_fixed_kernel<40>, grid, dim3(64), 0, as_stream(stream), stats_partial, group_size, M, eps, row_stats);
  else if (groups == 32)                 // DINOv2 ViT-L (1024 channels)
    hipLaunchKernelGGL(ln_stats_finalize_fixed_kernel<32>, grid, dim3(64), 0, as_stream(stream), stats_partial, group_size, M, eps, row_stats);
  else
    hipLaunchKernelGGL(ln_stats_finalize_kernel, grid, dim3(64), 0, as_stream(stream), stats_partial, groups, group_size, M, eps, row_stats);
  return launch_status();
}

extern "C" int s6d_row_stats_bf16(const void *x, long ldx, long M, int C, float eps, float *row_stats, void *stream) {
  if (M < 0 || C <= 0 || (C % 8) || ldx < C || (ldx % 8)) return S6D_EINVAL;
  if (C > 64 * 8 * 3) return S6D_EUNSUPPORTED;
  if (M == 0) return S6D_OK;
  if (!x || !row_stats || ((uintptr_t)x & 15) || ((uintptr_t)row_stats & 7)) return S6D_EINVAL;
  const dim3 grid((unsigned)((M + 3) / 4));
  const int vec = (C + 511) / 512;
  if (vec == 1)
    hipLaunchKernelGGL(row_stats_kernel<1>, grid, dim3(256), 0, as_stream(stream), (const u16 *)x, ldx, M, C, eps, row_stats);
  else if (vec == 2)
    hipLaunchKernelGGL(row_stats_kernel<2>, grid, dim3(256), 0, as_stream(stream), (const u16 *)x, ldx, M, C, eps, row_stats);
  else
    hipLaunchKernelGGL(row_stats_kernel<3>, grid, dim3(256), 0, as_stream(stream), (const u16 *)x, ldx, M, C, eps, row_stats);
  return launch_status();
}

static int ln_launch(const void *x, const void *delta, const float *gamma, const float *beta, float eps, long rows, int C,
                     void *x_out, void *y_out, float *y_f32, void *stream, bool f16 = false);

extern "C" int s6d_add_layernorm_f16(const void *x, const void *delta, const float *gamma, const float *beta,
                                     float eps, long rows, int C, void *x_out, void *y_out, void *stream) {
  if (!y_out) return S6D_EINVAL;
  return ln_launch(x, delta, gamma, beta, eps, rows, C, x_out, y_out, nullptr, stream, true);
}

extern "C" int s6d_add_layernorm_bf16(const void *x, const void *delta, const float *gamma, const float *beta,
                                      float eps, long rows, int C, void *x_out, void *y_out, void *stream) {
  if (!y_out) return S6D_EINVAL;
  return ln_launch(x, delta, gamma, beta, eps, rows, C, x_out, y_out, nullptr, stream);
}

extern "C" int s6d_layernorm_bf16_f32(const void *x, const float *gamma, const float *beta, float eps, long rows, int C,
                                      float *y_f32, void *stream) {
  if (!y_f32) return S6D_EINVAL;
  return ln_launch(x, nullptr, gamma, beta, eps, rows, C, nullptr, nullptr, y_f32, stream);
}

static int ln_launch(const void *x, const void *delta, const float *gamma, const float *beta, float eps, long rows, int C,
                     void *x_out, void *y_out, float *y_f32, void *stream, bool f16) {
  if (rows < 0 || C <= 0 || (C % 8) != 0) return S6D_EINVAL;
  if (rows == 0) return S6D_OK;
  if (!x || !gamma || !beta || (delta && !x_out)) return S6D_EINVAL;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int nchunk = C / 8;
  hipStream_t st = as_stream(stream);
#define S6D_LN(V)                                                                                              \
  do {                                                                                                         \
    if (f16)                                                                                                   \
      hipLaunchKernelGGL((add_layernorm_kernel<V, true>), dim3(grid), dim3(256), 0, st, (const u16 *)x, (const u16 *)delta, \
                         gamma, beta, eps, rows, C, (u16 *)x_out, (u16 *)y_out, y_f32);                        \
    else                                                                                                       \
      hipLaunchKernelGGL((add_layernorm_kernel<V, false>), dim3(grid), dim3(256), 0, st, (const u16 *)x, (const u16 *)delta, \
                         gamma, beta, eps, rows, C, (u16 *)x_out, (u16 *)y_out, y_f32);                        \
  } while (0)
  if (nchunk <= 64) S6D_LN(1);
  else if (nchunk <= 128) S6D_LN(2);
  else if (nchunk <= 192) S6D_LN(3);
  else if (nchunk <= 256) S6D_LN(4);
  else return S6D_EUNSUPPORTED;
#undef S6D_LN
  return launch_status();
}

extern "C" int s6d_add_layernorm_fp8(const void *x, const void *delta, const float *gamma, const float *beta, float eps, long rows,
                                     int C, void *x_out, void *y8, unsigned char *yscale, void *stream);

extern "C" int s6d_layernorm_fp8(const void *x, const float *gamma, const float *beta, float eps, long rows, int C, void *y8,
                                 unsigned char *yscale, void *stream) {
  return s6d_add_layernorm_fp8(x, nullptr, gamma, beta, eps, rows, C, nullptr, y8, yscale, stream);
}

extern "C" int s6d_add_layernorm_fp8(const void *x, const void *delta, const float *gamma, const float *beta, float eps, long rows,
                                     int C, void *x_out, void *y8, unsigned char *yscale, void *stream) {
  if (rows < 0 || C <= 0 || (C % 8) != 0) return S6D_EINVAL;
  if (rows == 0) return S6D_OK;
  if (!x || !gamma || !beta || !y8 || !yscale || (delta && !x_out)) return S6D_EINVAL;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  const int nchunk = C / 8;
  hipStream_t st = as_stream(stream);
#define S6D_LN8(V)                                                                                              \
  hipLaunchKernelGGL((layernorm_fp8_kernel<V>), dim3(grid), dim3(256), 0, st, (const u16 *)x, (const u16 *)delta, gamma, beta, \
                     eps, rows, C, (u16 *)x_out, (unsigned char *)y8, yscale)
  if (nchunk <= 64) S6D_LN8(1);
  else if (nchunk <= 128) S6D_LN8(2);
  else if (nchunk <= 192) S6D_LN8(3);
  else if (nchunk <= 256) S6D_LN8(4);
  else return S6D_EUNSUPPORTED;
#undef S6D_LN8
  return launch_status();
}
