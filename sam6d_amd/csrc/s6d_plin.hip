// Fused fp32 Linear of the PEM point transformer (gfx950):  y = LN( res + act( x W^T + b ) ), every stage optional.
//
// Reference (Pose_Estimation_Model/model/transformer.py): the 256-wide layers of the sparse and dense transformers --
//   :93-148   MultiHeadAttention   proj_q / proj_k / proj_v (256 -> 256)
//   :182-197  AttentionOutput      norm(x + squeeze(relu(expand(x))))   (256 -> 512 -> 256)
//   :200-224  AttentionLayer       norm(linear(attention) + x)
//   :352-406  RPEMultiHeadAttention proj_q / k / v / p
//   :518-608  LinearAttention(-Layer) proj_q / k / v, linear + residual + norm, AttentionOutput
// plus in_proj / out_proj of coarse_point_matching.py / fine_point_matching.py.  The reference runs each as an fp32 cuBLAS GEMM
// followed by separate bias / ReLU / add / LayerNorm kernels; round 2 of this repo did the same with hipBLASLt + ATen (8.8 % of
// the step's GPU time outside this library).
//
// Arithmetic: fp32-class products on the bf16 matrix cores by a 3-term split (x = x_hi + x_lo, w = w_hi + w_lo in bf16;
// x_hi w_hi + x_lo w_hi + x_hi w_lo, fp32 accumulation: ~2^-17 relative per product; the same scheme as geo_embed_kernel and
// fine_sweep_kernel), bias / ReLU / residual / LayerNorm in fp32 registers, two-pass LayerNorm statistics.
//
// Structure: 256 threads, one 64-row x 256-column output tile per workgroup (grid.y walks N in steps of 256); wave w owns
// columns [64 w, 64 w + 64) of the tile for all 64 rows as 2 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16, product transposed:
// the W fragment is the A operand, so a lane owns one output row -- what the row-wise epilogue wants).  K is walked in chunks
// of 32: x rows are split hi / lo on the way into LDS, W arrives pre-split (bf16 hi, lo matrices made once per weight version
// by the host wrapper); the next chunk's global loads are in flight while the current one is multiplied; 50 KB of LDS per
// workgroup, so three workgroups share a CU and fill each other's barriers.
#include "s6d_common.h"
#include "s6d_plin_math.h"

namespace s6d {

typedef __attribute__((ext_vector_type(8))) __bf16 pl_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pl_f32x16;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) unsigned pl_u32x4;
typedef __attribute__((ext_vector_type(4))) float pl_f32x4;

constexpr int PL_ROWS = 64, PL_COLS = 256, PL_KC = 32, PL_STRIDE = 40;   // LDS row: 32 bf16 + 8 pad (80 B)
constexpr int PL_OFF_XH = 0, PL_OFF_XL = PL_OFF_XH + PL_ROWS * PL_STRIDE, PL_OFF_WH = PL_OFF_XL + PL_ROWS * PL_STRIDE,
              PL_OFF_WL = PL_OFF_WH + PL_COLS * PL_STRIDE, PL_LDS_ELEMS = PL_OFF_WL + PL_COLS * PL_STRIDE;   // 25600 elems = 50 KB
// rows whose (row >> 2 ^ row >> 3) is odd keep their 16-byte chunk pairs swapped: with an 80-byte row stride the two row sets
// of a ds_read_b128 lane group otherwise meet on 3 of 16 bank slots (same fix as geo_embed_kernel)
__device__ __forceinline__ int pl_swz(int row) { return ((row >> 2) ^ (row >> 3)) & 1; }

__device__ __forceinline__ void pl_split(float x, u16 &hi, u16 &lo) {
  union { __bf16 b; u16 u; } h, l;
  h.b = (__bf16)x;
  const float xh = __uint_as_float(((unsigned)h.u) << 16);
  l.b = (__bf16)(x - xh);
  hi = h.u;
  lo = l.u;
}

struct PlinParams {
  const float *x;
  long ldx;
  const u16 *wh, *wl;       // (N,K) bf16 hi / lo parts of W
  const float *bias;        // (N) or null
  const float *res;         // (M,N) or null, row stride ldr
  long ldr;
  const float *gamma, *beta;   // LayerNorm over the N = 256 outputs, or null
  float eps;
  float *y;
  long ldy;
  int M, N, K, act;
};

__global__ __launch_bounds__(256) void plin_kernel(PlinParams p) {
  __shared__ __attribute__((aligned(16))) u16 lds[PL_LDS_ELEMS];
  __shared__ float red[PL_ROWS][8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * PL_ROWS, n0 = blockIdx.y * PL_COLS;

  // ---- staging geometry.  x: thread -> row tid >> 2, 8 floats at 8 (tid & 3);  W: thread -> rows (tid >> 2) + 64 j, chunk tid & 3
  const int srow = tid >> 2, sq = tid & 3;
  const float *xsrc = p.x + (size_t)min(m0 + srow, p.M - 1) * p.ldx + sq * 8;
  const u16 *whsrc = p.wh + (size_t)(n0 + srow) * p.K + sq * 8, *wlsrc = p.wl + (size_t)(n0 + srow) * p.K + sq * 8;
  pl_f32x4 xr0, xr1;
  pl_u32x4 wh0, wh1, wh2, wh3, wl0, wl1, wl2, wl3;
  const size_t wj = (size_t)64 * p.K;
#define PL_GLOAD(kc)                                                                   \
  do {                                                                                 \
    xr0 = *reinterpret_cast<const pl_f32x4 *>(xsrc + (kc) * PL_KC);                    \
    xr1 = *reinterpret_cast<const pl_f32x4 *>(xsrc + (kc) * PL_KC + 4);                \
    wh0 = *reinterpret_cast<const pl_u32x4 *>(whsrc + (kc) * PL_KC);                   \
    wh1 = *reinterpret_cast<const pl_u32x4 *>(whsrc + wj + (kc) * PL_KC);              \
    wh2 = *reinterpret_cast<const pl_u32x4 *>(whsrc + 2 * wj + (kc) * PL_KC);          \
    wh3 = *reinterpret_cast<const pl_u32x4 *>(whsrc + 3 * wj + (kc) * PL_KC);          \
    wl0 = *reinterpret_cast<const pl_u32x4 *>(wlsrc + (kc) * PL_KC);                   \
    wl1 = *reinterpret_cast<const pl_u32x4 *>(wlsrc + wj + (kc) * PL_KC);              \
    wl2 = *reinterpret_cast<const pl_u32x4 *>(wlsrc + 2 * wj + (kc) * PL_KC);          \
    wl3 = *reinterpret_cast<const pl_u32x4 *>(wlsrc + 3 * wj + (kc) * PL_KC);          \
  } while (0)
  const int ox = srow * PL_STRIDE + ((sq ^ pl_swz(srow)) * 8);
  int ow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ow[j] = (srow + 64 * j) * PL_STRIDE + ((sq ^ pl_swz(srow + 64 * j)) * 8);
  auto pack2 = [](u16 a, u16 b) -> unsigned { return (unsigned)a | ((unsigned)b << 16); };
#define PL_LSTORE()                                                                    \
  do {                                                                                 \
    u16 h[8], l[8];                                                                    \
    pl_split(xr0[0], h[0], l[0]); pl_split(xr0[1], h[1], l[1]); pl_split(xr0[2], h[2], l[2]); pl_split(xr0[3], h[3], l[3]); \
    pl_split(xr1[0], h[4], l[4]); pl_split(xr1[1], h[5], l[5]); pl_split(xr1[2], h[6], l[6]); pl_split(xr1[3], h[7], l[7]); \
    pl_u32x4 vh = {pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])};                           \
    pl_u32x4 vl = {pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])};                           \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_XH + ox) = vh;                          \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_XL + ox) = vl;                          \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WH + ow[0]) = wh0;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WH + ow[1]) = wh1;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WH + ow[2]) = wh2;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WH + ow[3]) = wh3;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WL + ow[0]) = wl0;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WL + ow[1]) = wl1;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WL + ow[2]) = wl2;                      \
    *reinterpret_cast<pl_u32x4 *>(lds + PL_OFF_WL + ow[3]) = wl3;                      \
  } while (0)

  // ---- fragments: lane -> row lane & 31 of the 32-row tile, 8 consecutive k at 16 ks + 8 (lane >> 5)
  const int fr = lane & 31, fh = lane >> 5;
  auto frag = [&](int off, int row, int ks) __attribute__((always_inline)) -> pl_bf16x8 {
    return *reinterpret_cast<const pl_bf16x8 *>(lds + off + row * PL_STRIDE + (((2 * ks + fh) ^ pl_swz(row)) * 8));
  };

  pl_f32x16 acc[2][2];                       // [m tile][n tile]; lane -> x row 32 mt + (lane & 31); register r -> output column
#pragma unroll                               //   64 wave + 32 nt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nkc = p.K / PL_KC;
  PL_GLOAD(0);
  for (int kc = 0; kc < nkc; ++kc) {
    __syncthreads();                         // the previous chunk's fragments have been read
    PL_LSTORE();
    __syncthreads();
    if (kc + 1 < nkc) PL_GLOAD(kc + 1);      // in flight under the products below
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      pl_bf16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xh[t] = frag(PL_OFF_XH, 32 * t + fr, ks);
        xl[t] = frag(PL_OFF_XL, 32 * t + fr, ks);
        wh[t] = frag(PL_OFF_WH, 64 * wave + 32 * t + fr, ks);
        wl[t] = frag(PL_OFF_WL, 64 * wave + 32 * t + fr, ks);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[nt], xh[mt], acc[mt][nt], 0, 0, 0);   // small terms first
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], xl[mt], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], xh[mt], acc[mt][nt], 0, 0, 0);
        }
    }
  }

#undef PL_GLOAD
#undef PL_LSTORE
  // ---- epilogue in registers: + bias, activation, + residual, LayerNorm over the row
  const int cb = n0 + 64 * wave + 4 * fh;    // column of register r of n tile nt: cb + 32 nt + (r & 3) + 8 (r >> 2)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b = *reinterpret_cast<const float4 *>(p.bias + cb + 32 * nt + 8 * q);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = min(m0 + 32 * mt + fr, p.M - 1);
        float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res) rs = *reinterpret_cast<const float4 *>(p.res + (size_t)m * p.ldr + cb + 32 * nt + 8 * q);
        const float rv[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][nt][4 * q + e] = pl_bias_act_res(acc[mt][nt][4 * q + e], bv[e], p.act == 1, rv[e]);
      }
    }
  if (p.gamma) {                             // N == 256 (launcher): the workgroup holds whole rows; statistics in two passes
    float mean[2], rstd[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s = pass ? pl_sqdev(s, acc[mt][nt][r], mean[mt]) : s + acc[mt][nt][r];
        red[32 * mt + fr][2 * wave + fh] = s;
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[32 * mt + fr][j];      // fixed order: every lane of a row gets the same value
        if (pass) rstd[mt] = rsqrtf(s / (float)PL_COLS + p.eps);
        else mean[mt] = s / (float)PL_COLS;
      }
      __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g = *reinterpret_cast<const float4 *>(p.gamma + cb + 32 * nt + 8 * q);
        const float4 bt = *reinterpret_cast<const float4 *>(p.beta + cb + 32 * nt + 8 * q);
        const float gv[4] = {g.x, g.y, g.z, g.w}, tv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][nt][4 * q + e] = pl_normalize(acc[mt][nt][4 * q + e], mean[mt], rstd[mt], gv[e], tv[e]);
      }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + 32 * mt + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(p.y + (size_t)m * p.ldy + cb + 32 * nt + 8 * q) =
            make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
  }
}

// W (N,K) fp32 -> bf16 hi / lo parts (once per weight version)
__global__ void plin_split_kernel(const float *__restrict__ w, long n, u16 *__restrict__ hi, u16 *__restrict__ lo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pl_split(w[i], hi[i], lo[i]);
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_linear_split_weight_f32(const float *w, long n, void *hi, void *lo, void *stream) {
  if (n < 0) return S6D_EINVAL;
  if (n == 0) return S6D_OK;
  if (!w || !hi || !lo) return S6D_EINVAL;
  hipLaunchKernelGGL(plin_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, n, (u16 *)hi, (u16 *)lo);
  return launch_status();
}

extern "C" int s6d_linear_f32(const float *x, long ldx, int M, int K, const void *w_hi, const void *w_lo, const float *bias, int N,
                              int act, const float *res, long ldr, const float *gamma, const float *beta, float eps, float *y,
                              long ldy, void *stream) {
  if (M < 0 || K <= 0 || N <= 0 || (act != 0 && act != 1)) return S6D_EINVAL;
  if (K % PL_KC != 0 || N % PL_COLS != 0) return S6D_EUNSUPPORTED;
  if ((gamma != nullptr) != (beta != nullptr)) return S6D_EINVAL;
  if (gamma && N != PL_COLS) return S6D_EUNSUPPORTED;                  // LayerNorm needs the whole row in one workgroup
  if (M == 0) return S6D_OK;
  if (!x || !w_hi || !w_lo || !y || ldx < K || ldy < N || (res && ldr < N)) return S6D_EINVAL;
  if ((ldx % 4) || (ldy % 4) || (res && (ldr % 4))) return S6D_EINVAL;  // float4 rows
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_hi | (uintptr_t)w_lo | (uintptr_t)res | (uintptr_t)bias | (uintptr_t)gamma |
       (uintptr_t)beta) & 15)
    return S6D_EINVAL;
  PlinParams p;
  p.x = x; p.ldx = ldx; p.wh = (const u16 *)w_hi; p.wl = (const u16 *)w_lo; p.bias = bias; p.res = res; p.ldr = ldr;
  p.gamma = gamma; p.beta = beta; p.eps = eps; p.y = y; p.ldy = ldy; p.M = M; p.N = N; p.K = K; p.act = act;
  hipLaunchKernelGGL(plin_kernel, dim3((unsigned)((M + PL_ROWS - 1) / PL_ROWS), (unsigned)(N / PL_COLS)), dim3(256), 0,
                     as_stream(stream), p);
  return launch_status();
}
