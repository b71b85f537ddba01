// PositionalEncoding of the fine point matching, fused (gfx950).
//
// Reference: PositionalEncoding.forward, Pose_Estimation_Model/model/fine_point_matching.py:101-125:
//   QueryAndGroup (ball query + 2 x group_points, pointnet2_utils.py:294-376) -> (B,6,N,ns)
//   SharedMLP [6,32,64,128] = 3 x (1x1 conv + BatchNorm + ReLU) (pytorch_utils.py:25-50) -> (B,128,N,ns): 67 MB/instance
//   max over the ns neighbours.
// Here one wavefront walks a point's neighbourhood in tiles of 16 neighbours and keeps the whole 6->32->64->128
// chain in registers: layer l's output tile in MFMA C layout (feature = g*4+r, neighbour = lane&15) IS layer
// l+1's B operand, with the weight matrix as the A operand.  BatchNorm (eval) is folded into the weights by the caller;
// bias + ReLU commute with the neighbour max, so the last layer only keeps a running max of raw accumulators.
// Nothing but the (B,N,128) result reaches HBM.
//
// Arithmetic (round 5): layers 1 and 2 on the bf16 matrix cores by the 3-term split of plin_kernel / geo_embed_kernel /
// fine_sweep_kernel (w = w_hi + w_lo, h = h_hi + h_lo in bf16; w_lo h_hi + w_hi h_lo + w_hi h_hi with fp32 accumulation:
// ~2^-17 relative per product) on v_mfma_f32_16x16x32_bf16 -- 6 matrix instructions of 16 cycles per 16 x 16 output tile of
// layer 2 where the exact-fp32 form of rounds 1-4 (v_mfma_f32_16x16x4_f32) needed 16 of 32 cycles: 960 instead of 5120 matrix
// cycles per 16 neighbours.  The k slot (g, e) of a matrix instruction carries feature 4 g + e (e < 4) / 16 + 4 g + e - 4 of
// its 32-feature block -- exactly the eight values the previous layer's C tiles leave in this lane -- and the weights are
// staged in LDS in that order (split once per workgroup), one conflict-free 16-byte read per lane and operand.
// Layer 0 (K = 6) stays fp32 VALU.
#include "s6d_common.h"

namespace s6d {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 pe_bf16x8;
typedef unsigned short u16;

constexpr int PE_WAVES = 4;

__device__ __forceinline__ void pe_split(float x, u16 &hi, u16 &lo) {       // pl_split of csrc/s6d_plin.hip
  union { __bf16 b; u16 u; } h, l;
  h.b = (__bf16)x;
  const float xh = __uint_as_float(((unsigned)h.u) << 16);
  l.b = (__bf16)(x - xh);
  hi = h.u;
  lo = l.u;
}
// the eight values of a lane's k slots -> hi / lo operands
__device__ __forceinline__ void pe_split8(const f32x4 &a, const f32x4 &b, pe_bf16x8 &hi, pe_bf16x8 &lo) {
  union { pe_bf16x8 v; u16 h[8]; } uh, ul;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    pe_split(a[e], uh.h[e], ul.h[e]);
    pe_split(b[e], uh.h[4 + e], ul.h[4 + e]);
  }
  hi = uh.v;
  lo = ul.v;
}

// LDS image of the split weights: operand (tile, k block, part) = 64 lanes x 16 B, lane-linear; part 0 = hi, 1 = lo
constexpr int PE_W1_OPS = 4 * 2, PE_W2_OPS = 8 * 2 * 2;                        // [u][part], [v][kb][part]

__global__ __launch_bounds__(PE_WAVES * 64, 4) void pe_group_mlp_kernel(const float *__restrict__ pts, const int32_t *__restrict__ idx,
                                                                   int B, int N, int ns, const float *__restrict__ W0,
                                                                   const float *__restrict__ b0, const float *__restrict__ W1,
                                                                   const float *__restrict__ b1, const float *__restrict__ W2,
                                                                   const float *__restrict__ b2, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) u16 sW1[PE_W1_OPS * 64 * 8];         // 8 KB
  __shared__ __attribute__((aligned(16))) u16 sW2[PE_W2_OPS * 64 * 8];         // 32 KB
  __shared__ float sW0[32 * 6], sb0[32], sb1[64], sb2[128];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  // operand element (tile, kb, lane (c, g), e): weight row 16 tile + c, column 32 kb + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)
  for (int i = tid; i < 4 * 64 * 8; i += PE_WAVES * 64) {
    const int e = i & 7, ln = (i >> 3) & 63, u = i >> 9;
    const int col = (e < 4) ? 4 * (ln >> 4) + e : 16 + 4 * (ln >> 4) + (e - 4);
    u16 hi, lo;
    pe_split(W1[(16 * u + (ln & 15)) * 32 + col], hi, lo);
    sW1[((u * 2 + 0) * 64 + ln) * 8 + e] = hi;
    sW1[((u * 2 + 1) * 64 + ln) * 8 + e] = lo;
  }
  for (int i = tid; i < 16 * 64 * 8; i += PE_WAVES * 64) {
    const int e = i & 7, ln = (i >> 3) & 63, vk = i >> 9, v = vk >> 1, kb = vk & 1;
    const int col = 32 * kb + ((e < 4) ? 4 * (ln >> 4) + e : 16 + 4 * (ln >> 4) + (e - 4));
    u16 hi, lo;
    pe_split(W2[(16 * v + (ln & 15)) * 64 + col], hi, lo);
    sW2[((vk * 2 + 0) * 64 + ln) * 8 + e] = hi;
    sW2[((vk * 2 + 1) * 64 + ln) * 8 + e] = lo;
  }
  for (int i = tid; i < 32 * 6; i += PE_WAVES * 64) sW0[i] = W0[i];
  if (tid < 32) sb0[tid] = b0[tid];
  if (tid < 64) sb1[tid] = b1[tid];
  if (tid < 128) sb2[tid] = b2[tid];
  __syncthreads();
  auto w1op = [&](int u, int part) __attribute__((always_inline)) -> pe_bf16x8 {
    return *reinterpret_cast<const pe_bf16x8 *>(sW1 + ((u * 2 + part) * 64 + lane) * 8);
  };
  auto w2op = [&](int v, int kb, int part) __attribute__((always_inline)) -> pe_bf16x8 {
    return *reinterpret_cast<const pe_bf16x8 *>(sW2 + (((v * 2 + kb) * 2 + part) * 64 + lane) * 8);
  };

  const long npts = (long)B * N;
  const int ntile = (ns + 15) / 16;
  for (long pt = (long)blockIdx.x * PE_WAVES + wave; pt < npts; pt += (long)gridDim.x * PE_WAVES) {
    const int b = (int)(pt / N);
    const float *P = pts + (size_t)b * N * 3;
    const float cx = pts[pt * 3], cy = pts[pt * 3 + 1], cz = pts[pt * 3 + 2];
    f32x4 mx[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) mx[v] = f32x4{-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
    for (int tt = 0; tt < ntile; ++tt) {
#ifndef HIPEMU
      asm volatile("" ::: "memory");     // the weight operands are READ per tile (40 KB of LDS, 40 reads): hoisted they are 160 registers
#endif
      const int k = min(tt * 16 + c, ns - 1);                      // surplus lanes repeat the last neighbour (max unchanged)
      const int j = idx[pt * ns + k];
      const float nx = P[j * 3], ny = P[j * 3 + 1], nz = P[j * 3 + 2];
      const float x[6] = {nx - cx, ny - cy, nz - cz, nx, ny, nz};   // [grouped_xyz - centre ; grouped features = xyz]
      // layer 0 (VALU, fp32): features f = 16t + g*4 + r of neighbour c
      f32x4 h0[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + g * 4 + r;
          float a = sb0[f];
#pragma unroll
          for (int d = 0; d < 6; ++d) a += sW0[f * 6 + d] * x[d];
          h0[t][r] = fmaxf(a, 0.f);
        }
      // layer 1: H1^T (64 x 16) = W1 (64 x 32) H0^T in one k block; the lane's eight k slots are h0[0][0..3], h0[1][0..3]
      pe_bf16x8 xh, xl;
      pe_split8(h0[0], h0[1], xh, xl);
      f32x4 h1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const pe_bf16x8 wh = w1op(u, 0), wl = w1op(u, 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc, 0, 0, 0);                  // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[u][r] = fmaxf(acc[r] + sb1[16 * u + g * 4 + r], 0.f);
      }
      // layer 2: H2^T (128 x 16) = W2 (128 x 64) H1^T in two k blocks (tiles 2 kb, 2 kb + 1 of h1); running max of the raw accumulators
      pe_bf16x8 yh[2], yl[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) pe_split8(h1[2 * kb], h1[2 * kb + 1], yh[kb], yl[kb]);
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const pe_bf16x8 wh = w2op(v, kb, 0), wl = w2op(v, kb, 1);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, yh[kb], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, yl[kb], acc, 0, 0, 0);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2op(v, kb, 0), yh[kb], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) mx[v][r] = fmaxf(mx[v][r], acc[r]);
      }
    }
    // max over the 16 neighbour lanes, then + bias, ReLU (both commute with the max)
#pragma unroll
    for (int v = 0; v < 8; ++v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m = mx[v][r];
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        m = fmaxf(m, __shfl_xor(m, 4));
        m = fmaxf(m, __shfl_xor(m, 8));
        mx[v][r] = fmaxf(m + sb2[16 * v + g * 4 + r], 0.f);
      }
      if (c == 0) *reinterpret_cast<float4 *>(out + pt * 128 + 16 * v + g * 4) = make_float4(mx[v][0], mx[v][1], mx[v][2], mx[v][3]);
    }
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_pe_group_mlp_f32(const float *pts, const int32_t *idx, int B, int N, int ns, const float *W0,
                                    const float *b0, const float *W1, const float *b1, const float *W2, const float *b2,
                                    float *out, void *stream) {
  if (B < 0 || N <= 0 || ns <= 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!pts || !idx || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !out) return S6D_EINVAL;
  const long npts = (long)B * N;
  long grid = (npts + PE_WAVES - 1) / PE_WAVES;
  if (grid > 256 * 8) grid = 256 * 8;                             // persistent-ish: weights are staged once per workgroup
  hipLaunchKernelGGL(pe_group_mlp_kernel, dim3((unsigned)grid), dim3(PE_WAVES * 64), 0, as_stream(stream), pts, idx, B, N, ns,
                     W0, b0, W1, b1, W2, b2, out);
  return launch_status();
}
