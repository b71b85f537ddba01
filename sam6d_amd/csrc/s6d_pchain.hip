// The post-attention chain of a PEM transformer layer as ONE kernel (gfx950; round 6, VERDICT r5 next #4 i):
//
//     h = LayerNorm1( x + a W1^T + b1 )                      AttentionLayer / RPEAttentionLayer / LinearAttentionLayer: linear + residual + norm
//     y = LayerNorm2( h + relu(h We^T + be) Ws^T + bs )      AttentionOutput: expand 256 -> 512, ReLU, squeeze 512 -> 256, residual, norm
//
// Reference: Pose_Estimation_Model/model/transformer.py:182-197 (AttentionOutput), :200-224 (AttentionLayer), :409-438
// (RPEAttentionLayer), :567-608 (LinearAttentionLayer): four nn.Linear / LayerNorm statements per layer, each its own cuBLAS /
// elementwise launch there.  Rounds 3-5 ran them as three launches of plin_kernel (csrc/s6d_plin.hip); h and the 512-wide
// activations went through HBM between them (470 MB per layer at 32 instances x 2048 points) and the 197-token layers paid
// three launch floors.  Here a workgroup keeps a 32-row strip on chip from a to y.
//
// Arithmetic = plin_kernel's, operation for operation (3-term bf16 split on v_mfma_f32_32x32x16_bf16, small terms first, K walked
// in chunks of 32 in ascending order, bias / ReLU / residual / two-pass LayerNorm in fp32 with the same fixed-order row
// reductions), so the result equals the three-launch form BIT FOR BIT (tests/test_gpu_plin.py, tests/test_emu_plin.py) and the
// PEM goldens and batch-invariance tests do not move.
//
// Structure: 256 threads, 64 rows per workgroup, one workgroup per CU (137 KB of LDS, 502 registers per lane, no spills).
//   * wave w owns columns [64 w, 64 w + 64) of every 256-wide product (2 x 2 MFMA tiles, product transposed: a lane owns one row).
//     Its W rows are therefore PRIVATE to it, and W never touches LDS: the host keeps a copy of every weight in FRAGMENT ORDER
//     (s6d_linear_fragment_weight: for each 32-row tile and each 16-wide k step the 64 lanes' 16-byte fragments back to back, 1 KiB)
//     and a lane's A operand is one coalesced 16-byte load, three chunks (96 k) ahead in registers.  The ring of fragment registers
//     lives ACROSS the stages: the first three chunks of a product's W are requested before the previous stage's epilogue
//     (bias / LayerNorm / image writes / barriers), so the L2 round trip is never in front of the first matrix instruction.
//     `__builtin_amdgcn_sched_barrier(0)` after every group of loads: without it the scheduler sinks each load to its first use
//     (measured: 113.8 us average per launch with the loads sunk, the PEM stage 2 % slower than three launches; with the ring kept,
//     the stage is 5.6 % faster at 32 instances -- 22.10 -> 20.87 ms -- and 7 % at 10 -- 12.45 -> 11.55 ms).
//     (First form of this kernel: 32-row strips with plin_kernel's W chunk staged through LDS behind two barriers per chunk -- 324 us
//     at 65536 rows, no faster than the three launches: 8 ds_write_b128 per thread and chunk for 12 MFMAs.)
//   * the B operands are full-row images in LDS in split (hi / lo) form, written once per stage: A (the attention output strip),
//     H (h) and E (one 256-column half of the expanded activations; it reuses A's buffer).  Rows are 528 bytes apart: every
//     ds_read_b128 lane group covers 16 distinct bank slots.  The 512-wide activation never exists as a whole: for each half,
//     expand -> ReLU -> E -> 256 more k of the squeeze product.  h stays in registers as the second residual.
//   * no barrier inside a product; six per strip between the stages, plus the LayerNorm reductions.
#include "s6d_common.h"
#include "s6d_plin_math.h"

namespace s6d {

typedef __attribute__((ext_vector_type(8))) __bf16 pc_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pc_f32x16;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) unsigned pc_u32x4;
typedef __attribute__((ext_vector_type(4))) float pc_f32x4;

constexpr int PC_FS = 264;                                 // image row: 256 bf16 + 8 pad (528 B)
// MT = 32-row tiles per workgroup: 2 (64 rows, 137 KB of LDS) where the rows fill the chip, 1 (32 rows, 69 KB) up to 8192 rows --
// the 197-token layers (6304 rows at 32 instances, 1970 at 10) then run on twice the CUs with half the serial work per workgroup
// (334 registers per lane: still one workgroup per CU, which is all such a launch has).  A row's arithmetic does not depend on
// MT: the same bits.
template <int MT>
struct PcCfg {
  static constexpr int ROWS = 32 * MT;
  static constexpr int IMG = ROWS * PC_FS;                 // elements of one (hi or lo) image
  static constexpr int OFF_AH = 0, OFF_AL = IMG, OFF_HH = 2 * IMG, OFF_HL = 3 * IMG, LDS_ELEMS = 4 * IMG;
  static constexpr int LDS_BYTES = LDS_ELEMS * 2 + ROWS * 8 * 4;
};

__device__ __forceinline__ void pc_split(float x, u16 &hi, u16 &lo) {
  union { __bf16 b; u16 u; } h, l;
  h.b = (__bf16)x;
  const float xh = __uint_as_float(((unsigned)h.u) << 16);
  l.b = (__bf16)(x - xh);
  hi = h.u;
  lo = l.u;
}

struct PchainParams {
  const float *a;            // (M,256) attention output, row stride lda
  long lda;
  const float *x;            // (M,256) the layer's input (first residual), row stride ldx
  long ldx;
  const u16 *w1h, *w1l;      // (256,256) linear, bf16 hi / lo, FRAGMENT ORDER
  const float *b1, *g1, *be1;
  float eps1;
  const u16 *weh, *wel;      // (512,256) expand
  const float *bexp;
  const u16 *wsh, *wsl;      // (256,512) squeeze
  const float *bsq, *g2, *be2;
  float eps2;
  float *y;
  long ldy;
  int M;
};

extern __shared__ __attribute__((aligned(16))) char pc_smem[];

template <int MT>
__global__ __launch_bounds__(256) void pchain_kernel(PchainParams p) {
  using Cf = PcCfg<MT>;
  constexpr int PC_ROWS = Cf::ROWS, PC_OFF_AH = Cf::OFF_AH, PC_OFF_AL = Cf::OFF_AL, PC_OFF_HH = Cf::OFF_HH, PC_OFF_HL = Cf::OFF_HL, PC_LDS_ELEMS = Cf::LDS_ELEMS;
  u16 *lds = reinterpret_cast<u16 *>(pc_smem);
  float(*red)[8] = reinterpret_cast<float(*)[8]>(pc_smem + PC_LDS_ELEMS * 2);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * PC_ROWS;
  const int fr = lane & 31, fh = lane >> 5;
  auto pack2 = [](u16 a, u16 b) -> unsigned { return (unsigned)a | ((unsigned)b << 16); };

  // One 64 x 256 product over K = 256:  acc[mt][nt] += X W^T.  X = the split image at (xh_off, xl_off), columns [0, 256).  W = the
  // 64 rows [r0 + 64 wave, + 64) of a weight in fragment order with k16 steps per row tile (K / 16), k steps [k16_0, k16_0 + 16).
  // `prefetch` issues the first PC_PF chunks of a product's W fragments; it is called BEFORE the epilogue of the previous stage so
  // that the L2 round trip runs under the LayerNorm / image writes / barriers instead of in front of the first matrix instruction.
  constexpr int NKC = 8, PC_PF = 3;
  pc_u32x4 wr[PC_PF][8];                                     // [chunk slot][hi: nt 0 ks 0, nt 0 ks 1, nt 1 ks 0, nt 1 ks 1 | lo: the same]
  auto gload = [&](const u16 *wh, const u16 *wl, size_t t0, size_t tn, int kc, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const size_t o = t0 + nt * tn + (size_t)(2 * kc + ks) * 512;
        wr[s][2 * nt + ks] = *reinterpret_cast<const pc_u32x4 *>(wh + o);
        wr[s][4 + 2 * nt + ks] = *reinterpret_cast<const pc_u32x4 *>(wl + o);
      }
  };
  auto prefetch = [&](const u16 *wh, const u16 *wl, int k16, int r0, int k16_0) __attribute__((always_inline)) {
    const size_t t0 = ((size_t)((r0 >> 5) + 2 * wave) * k16 + k16_0) * 512 + lane * 8;        // n tile 0, first k step, this lane
#pragma unroll
    for (int s = 0; s < PC_PF; ++s) gload(wh, wl, t0, (size_t)k16 * 512, s, s);
    __builtin_amdgcn_sched_barrier(0);                       // (the scheduler otherwise sinks every load to its first use: no prefetch)
  };
  auto gemm = [&](pc_f32x16 (&acc)[MT][2], const u16 *wh, const u16 *wl, int k16, int r0, int k16_0, int xh_off, int xl_off)
      __attribute__((always_inline)) {
    const size_t t0 = ((size_t)((r0 >> 5) + 2 * wave) * k16 + k16_0) * 512 + lane * 8;
    const size_t tn = (size_t)k16 * 512;                                                       // to the next n tile
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      const int s = kc % PC_PF;
      pc_bf16x8 whf[2][2], wlf[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          whf[nt][ks] = __builtin_bit_cast(pc_bf16x8, wr[s][2 * nt + ks]);
          wlf[nt][ks] = __builtin_bit_cast(pc_bf16x8, wr[s][4 + 2 * nt + ks]);
        }
      if (kc + PC_PF < NKC) gload(wh, wl, t0, tn, kc + PC_PF, s);   // in flight under the products of this and the next three chunks
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        pc_bf16x8 xh[MT], xl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int o = (32 * mt + fr) * PC_FS + kc * 32 + (2 * ks + fh) * 8;
          xh[mt] = *reinterpret_cast<const pc_bf16x8 *>(lds + xh_off + o);
          xl[mt] = *reinterpret_cast<const pc_bf16x8 *>(lds + xl_off + o);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlf[nt][ks], xh[mt], acc[mt][nt], 0, 0, 0);   // small terms first
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whf[nt][ks], xl[mt], acc[mt][nt], 0, 0, 0);   // (plin_kernel's order)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whf[nt][ks], xh[mt], acc[mt][nt], 0, 0, 0);
          }
      }
    }
  };
  const int cb = 64 * wave + 4 * fh;                        // column of register r of n tile nt: cb + 32 nt + (r & 3) + 8 (r >> 2)
  // row-wise LayerNorm of the 256 values a row's eight lanes hold, in plin_kernel's arithmetic (two passes, fixed-order sums)
  auto layernorm = [&](pc_f32x16 (&acc)[MT][2], const float *gamma, const float *beta, float eps) __attribute__((always_inline)) {
    float mean[MT], rstd[MT];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s = pass ? pl_sqdev(s, acc[mt][nt][r], mean[mt]) : s + acc[mt][nt][r];
        red[32 * mt + fr][2 * wave + fh] = s;
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[32 * mt + fr][j];
        if (pass) rstd[mt] = rsqrtf(t / 256.f + eps);
        else mean[mt] = t / 256.f;
      }
      __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g = *reinterpret_cast<const float4 *>(gamma + cb + 32 * nt + 8 * q);
        const float4 bt = *reinterpret_cast<const float4 *>(beta + cb + 32 * nt + 8 * q);
        const float gv[4] = {g.x, g.y, g.z, g.w}, tv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][nt][4 * q + e] = pl_normalize(acc[mt][nt][4 * q + e], mean[mt], rstd[mt], gv[e], tv[e]);
      }
  };
  // the values a lane holds -> the split operand image (rows 32 mt + fr, its columns)
  auto to_image = [&](const pc_f32x16 (&v)[MT][2], int off_hi, int off_lo) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u16 h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) pc_split(v[mt][nt][4 * q + e], h[e], l[e]);
          const int o = (32 * mt + fr) * PC_FS + cb + 32 * nt + 8 * q;
          *reinterpret_cast<uint2 *>(lds + off_hi + o) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
          *reinterpret_cast<uint2 *>(lds + off_lo + o) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
        }
  };
  auto zero = [](pc_f32x16 (&acc)[MT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  };

  // ---- the attention-output strip -> image A: thread -> row tid >> 2, 64 floats at 64 (tid & 3); rows past M repeat row M - 1
  prefetch(p.w1h, p.w1l, 16, 0, 0);
  {
    constexpr int TPR = 8 / MT, FPT = 32 * MT;               // threads per row, floats per thread
    const int row = tid / TPR, c0 = FPT * (tid % TPR);
    const float *src = p.a + (size_t)min(m0 + row, p.M - 1) * p.lda + c0;
    pc_f32x4 v[FPT / 4];
#pragma unroll
    for (int i = 0; i < FPT / 4; ++i) v[i] = *reinterpret_cast<const pc_f32x4 *>(src + 4 * i);
#pragma unroll
    for (int i = 0; i < FPT / 4; i += 2) {
      u16 h[8], l[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pc_split(v[i][e], h[e], l[e]);
        pc_split(v[i + 1][e], h[4 + e], l[4 + e]);
      }
      const int o = row * PC_FS + c0 + 4 * i;
      *reinterpret_cast<pc_u32x4 *>(lds + PC_OFF_AH + o) = (pc_u32x4){pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])};
      *reinterpret_cast<pc_u32x4 *>(lds + PC_OFF_AL + o) = (pc_u32x4){pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  // ---- stage 1: h = LN1(x + a W1^T + b1)
  pc_f32x16 h[MT][2];
  zero(h);
  gemm(h, p.w1h, p.w1l, 16, 0, 0, PC_OFF_AH, PC_OFF_AL);
  // the first residual: issued before the next product's W
  pc_f32x4 xres[MT][2][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xres[mt][nt][q] = *reinterpret_cast<const pc_f32x4 *>(p.x + (size_t)min(m0 + 32 * mt + fr, p.M - 1) * p.ldx + cb + 32 * nt + 8 * q);
  prefetch(p.weh, p.wel, 16, 0, 0);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4 *>(p.b1 + cb + 32 * nt + 8 * q);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) h[mt][nt][4 * q + e] = pl_bias_act_res(h[mt][nt][4 * q + e], bv[e], false, xres[mt][nt][q][e]);
    }
  layernorm(h, p.g1, p.be1, p.eps1);
  to_image(h, PC_OFF_HH, PC_OFF_HL);
  __syncthreads();                                          // H complete; every wave is past its reads of A (the reductions above)

  // ---- stages 2 + 3: y = LN2(h + relu(h We^T + be) Ws^T + bs), the 512 expanded columns in two halves (E lives in A's buffer)
  pc_f32x16 y[MT][2];
  zero(y);
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {
    pc_f32x16 e2[MT][2];
    zero(e2);
    gemm(e2, p.weh, p.wel, 16, 256 * c, 0, PC_OFF_HH, PC_OFF_HL);
    prefetch(p.wsh, p.wsl, 32, 0, 16 * c);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(p.bexp + 256 * c + cb + 32 * nt + 8 * q);
        const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) e2[mt][nt][4 * q + e] = pl_bias_act_res(e2[mt][nt][4 * q + e], bv[e], true, 0.f);
      }
    __syncthreads();                                        // every wave is past the previous half's squeeze product (reads of E)
    to_image(e2, PC_OFF_AH, PC_OFF_AL);
    __syncthreads();
    gemm(y, p.wsh, p.wsl, 32, 0, 16 * c, PC_OFF_AH, PC_OFF_AL);
    if (c == 0) prefetch(p.weh, p.wel, 16, 256, 0);
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4 *>(p.bsq + cb + 32 * nt + 8 * q);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[mt][nt][4 * q + e] = pl_bias_act_res(y[mt][nt][4 * q + e], bv[e], false, h[mt][nt][4 * q + e]);
    }
  layernorm(y, p.g2, p.be2, p.eps2);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 32 * mt + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(p.y + (size_t)m * p.ldy + cb + 32 * nt + 8 * q) =
            make_float4(y[mt][nt][4 * q], y[mt][nt][4 * q + 1], y[mt][nt][4 * q + 2], y[mt][nt][4 * q + 3]);
  }
}

// W (N,K) bf16 (one part of a split weight) -> fragment order: for n tile t (32 rows) and k step s (16 wide) the 64 lanes' 8-element
// fragments back to back: element (32 t + (lane & 31), 16 s + 8 (lane >> 5) + e) at ((t K/16 + s) 64 + lane) 8 + e.
__global__ void pchain_fragment_kernel(const u16 *__restrict__ w, int N, int K, u16 *__restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte fragment per thread
  const long total = (long)N * K / 8;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long ts = i >> 6;
  const int k16 = K / 16, s = (int)(ts % k16), t = (int)(ts / k16);
  const u16 *src = w + (size_t)(32 * t + (lane & 31)) * K + 16 * s + 8 * (lane >> 5);
  *reinterpret_cast<pc_u32x4 *>(out + i * 8) = *reinterpret_cast<const pc_u32x4 *>(src);
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_linear_fragment_weight(const void *w, int N, int K, void *out, void *stream) {
  if (N <= 0 || K <= 0 || (N % 32) || (K % 16)) return S6D_EINVAL;
  if (!w || !out || w == out) return S6D_EINVAL;
  const long total = (long)N * K / 8;
  hipLaunchKernelGGL(pchain_fragment_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), (const u16 *)w, N, K,
                     (u16 *)out);
  return launch_status();
}

extern "C" int s6d_attn_output_chain_f32(const float *a, long lda, const float *x, long ldx, int M, const void *w1_hi, const void *w1_lo,
                                         const float *b1, const float *gamma1, const float *beta1, float eps1, const void *we_hi,
                                         const void *we_lo, const float *be, const void *ws_hi, const void *ws_lo, const float *bs,
                                         const float *gamma2, const float *beta2, float eps2, float *y, long ldy, void *stream) {
  if (M < 0) return S6D_EINVAL;
  if (M == 0) return S6D_OK;
  if (!a || !x || !w1_hi || !w1_lo || !b1 || !gamma1 || !beta1 || !we_hi || !we_lo || !be || !ws_hi || !ws_lo || !bs || !gamma2 ||
      !beta2 || !y)
    return S6D_EINVAL;
  if (lda < 256 || ldx < 256 || ldy < 256 || (lda % 4) || (ldx % 4) || (ldy % 4)) return S6D_EINVAL;
  if (((uintptr_t)a | (uintptr_t)x | (uintptr_t)y) & 15) return S6D_EINVAL;
  PchainParams p;
  p.a = a; p.lda = lda; p.x = x; p.ldx = ldx; p.M = M;
  p.w1h = (const u16 *)w1_hi; p.w1l = (const u16 *)w1_lo; p.b1 = b1; p.g1 = gamma1; p.be1 = beta1; p.eps1 = eps1;
  p.weh = (const u16 *)we_hi; p.wel = (const u16 *)we_lo; p.bexp = be;
  p.wsh = (const u16 *)ws_hi; p.wsl = (const u16 *)ws_lo; p.bsq = bs; p.g2 = gamma2; p.be2 = beta2; p.eps2 = eps2;
  p.y = y; p.ldy = ldy;
  if (M <= 8192) {                                            // at most one 32-row workgroup per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pchain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, PcCfg<1>::LDS_BYTES);
    hipLaunchKernelGGL(pchain_kernel<1>, dim3((unsigned)((M + 31) / 32)), dim3(256), PcCfg<1>::LDS_BYTES, as_stream(stream), p);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pchain_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PcCfg<2>::LDS_BYTES);
    hipLaunchKernelGGL(pchain_kernel<2>, dim3((unsigned)((M + 63) / 64)), dim3(256), PcCfg<2>::LDS_BYTES, as_stream(stream), p);
  }
  return launch_status();
}
