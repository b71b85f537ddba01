// SAM two-way mask decoder, image-side kernels (gfx950) -- the per-prompt work of SURVEY.md section 8f-2.
//
// Reference: segment_anything/modeling/transformer.py TwoWayAttentionBlock.forward :151-182 step (4)
//     keys = norm4(keys + cross_attn_image_to_token(q = keys + key_pe, k = queries + query_pe, v = queries))
// and mask_decoder.py MaskDecoder.predict_masks :126-139
//     upscaled = output_upscaling(src) ; masks = hyper_in @ upscaled.view(b, c, h*w)
// As written both stream (B, 4096, 256) fp32 token tensors through ~10 elementwise / GEMM kernels per layer.
//
// img2tok_kernel: every image token attends to only n_tok <= 8 prompt tokens per head, so
//     out_proj(softmax(q k^T / 4) v) = softmax(q k^T / 4) (v W_o^T)
// is, per prompt, a (N x 64) x (64 x 256) product whose left factor is the softmax itself: scores by MFMA against
// a block-diagonal expansion of the 8 x 8 (head, token) keys, softmax in the accumulator layout (two lanes per
// (token, head): one DPP exchange), the probabilities re-used in place as the B operand of the value product,
// then + bias + residual and LayerNorm over the 256 channels in registers.  One read of q (and the residual), one
// write of the new keys: HBM-bound.
//
// upscale_heads_kernel: LayerNorm2d + GELU, the second 2x2/2 transposed conv (as a 64 -> 4x32 GEMM), GELU and the
// hypernetwork product per output pixel, from the first transposed conv's GEMM output: the (B, 65536, 32) upscaled
// embedding is never materialised.
#include "s6d_common.h"

namespace s6d {

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 sd_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sd_f32x4;

__device__ __forceinline__ float sd_bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ u16 sd_f2bf(float f) {
  union { __bf16 b; u16 u; } x;
  x.b = (__bf16)f;
  return x.u;
}

constexpr int kSdC = 256;        // embedding dim
constexpr int kSdD = 128;        // internal dim of the cross attentions (downsample rate 2)
constexpr int kSdJ = 64;         // (head, token slot) pairs: 8 heads x 8 slots
constexpr int kSdKRow = kSdD + 8;   // LDS row stride of the expanded keys (bf16): 272 B, conflict-free b128 reads
#ifndef S6D_SD_LDSFIX
#define S6D_SD_LDSFIX 1             // 0: the LDS layouts before round 6's conflict fixes (same-box A/B builds)
#endif
constexpr int kSdVRow = S6D_SD_LDSFIX ? kSdJ + 4 : kSdJ + 8;   // LDS row stride of the folded values (bf16): 136 B = 34 dwords -- sixteen consecutive rows start on
                                    // sixteen distinct even banks of the 32 a ds_read2_b64 access sees (round 6: with 144-byte rows
                                    // rows r and r + 8 met, and the permuted channel order of the same round made it four rows:
                                    // SQ_LDS_BANK_CONFLICT was 235 M of the kernel's 407 M LDS cycles per 1024 prompts; now 0, at
                                    // unchanged time: profiles/r06_samdec_ab.md)
// Fragment reads by ds_read_b128 from rows that are an ODD number of 16-byte slots long: a lane group is {rows 0-3, 12-15 at chunk g}
// + {rows 4-11 at chunk g + 1} (MI355X_MICROARCH.md, LDS), which meet on one slot (row 11's chunk g + 1 = row 12's chunk g); rows
// 4-11 therefore keep their chunk pairs swapped -- written and read at chunk ^ sd_kswz(row) (the swizzle of csrc/s6d_geo.hip /
// s6d_attn.hip, round 6 here: one extra LDS cycle per group on every K / W fragment read of the three kernels below).
__device__ __forceinline__ int sd_kswz(int row) { return S6D_SD_LDSFIX ? ((row >> 2) ^ (row >> 3)) & 1 : 0; }

// q (Bq,N,128) bf16 with row stride q_ld (+ q_add (N,128) bf16 or null), kexp (B,64,128) bf16: row j = h*8+t holds scale * k_t in the 16
// columns of head h, zeros elsewhere; vpt (B,256,64) bf16: vpt[n][j] = (v_t W_o^T)[n] restricted to head h;
// resid (Br,N,256) bf16; obias, gamma, beta (256) f32 -> out (B,N,256) bf16.  Bq, Br in {1, B}.
// RAW (round 4): the q projection is folded into the expanded keys -- q = the RAW image tokens (B|1,N,256), q_add = the positional
// encoding (N,256), kexp (B,64,256) = kexp_128 W_q, cbias (B,64) = kexp_128 . b_q (the projection's bias is NOT constant over the
// slots a token's softmax runs over): the (B,N,128) projected queries of the image tokens are never written or read.
template <bool RAW>
__global__ __launch_bounds__(256, 2) void img2tok_kernel(const u16 *__restrict__ q, const u16 *__restrict__ q_add,
                                                      const u16 *__restrict__ kexp, const float *__restrict__ cbias,
                                                      const u16 *__restrict__ vpt,
                                                      const u16 *__restrict__ resid, const float *__restrict__ obias,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      float eps, int N, int n_tok, int q_ld, long q_bstride,
                                                      long r_bstride, u16 *__restrict__ out) {
  constexpr int KD = RAW ? kSdC : kSdD, KROWL = KD + 8, KSN = KD / 32;   // key width, LDS row stride, k-steps of the score product
  // Output channel of row i of value tile nt (round 6).  In the accumulator layout lane (c, g) holds rows 4 g .. 4 g + 3 of every
  // tile for token c.  With channel = 16 nt + i a lane's four values of a tile were 8 bytes and a store / residual-load instruction
  // touched 32 contiguous bytes per token.  Feeding the value rows in the order chan(nt, i) = 32 (nt / 2) + 8 (i / 4) + 4 (nt % 2) + i % 4
  // makes the four values of tiles 2 p and 2 p + 1 eight consecutive channels: one 16-byte access per lane and tile pair, 64
  // contiguous bytes per token and instruction (half the instructions, twice the run length; same-box A/B: no difference in time,
  // profiles/r06_samdec_ab.md).  Per channel the arithmetic is
  // unchanged; the LayerNorm's per-lane partial sums run over another 64 channels, so the result can differ in the last float32 bit.
#ifndef S6D_SD_CHANPERM
#define S6D_SD_CHANPERM 1            // 0: channel = 16 nt + i with 8-byte residual loads / stores (rounds 2-5; same-box A/B builds)
#endif
  auto chan = [](int nt, int i) { return S6D_SD_CHANPERM ? 32 * (nt >> 1) + 8 * (i >> 2) + 4 * (nt & 1) + (i & 3) : 16 * nt + i; };
  extern __shared__ __attribute__((aligned(16))) char sd_smem[];
  u16 *Kl = reinterpret_cast<u16 *>(sd_smem);                       // [64][KROWL]
  u16 *Vl = Kl + kSdJ * KROWL;                                      // [256][kSdVRow]
  float *pl = reinterpret_cast<float *>(Vl + kSdC * kSdVRow);       // [3][256]: obias, gamma, beta
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < kSdJ * KD / 8; i += 256) {                  // 16-byte chunks
    const int row = i / (KD / 8), ch = i - row * (KD / 8);
    *reinterpret_cast<uint4 *>(Kl + row * KROWL + (ch ^ sd_kswz(row)) * 8) =
        *reinterpret_cast<const uint4 *>(kexp + ((size_t)b * kSdJ + row) * KD + ch * 8);
  }
  for (int i = tid; i < kSdC * kSdJ / 8; i += 256) {
    const int row = i / (kSdJ / 8), ch = i - row * (kSdJ / 8);
    // channel `row` goes to LDS row rho(row) = the position at which the value product reads it: tile nt = 2 (row / 32) + (row / 4 & 1),
    // row-in-tile i = 4 (row / 8 & 3) + row % 4 (the inverse of chan() below), so that a fragment read walks 16 CONSECUTIVE LDS rows
    const int rem = row & 31, lrow = (S6D_SD_LDSFIX && S6D_SD_CHANPERM) ? (2 * (row >> 5) + ((rem >> 2) & 1)) * 16 + 4 * (rem >> 3) + (rem & 3) : row;
    const uint4 v = *reinterpret_cast<const uint4 *>(vpt + ((size_t)b * kSdC + row) * kSdJ + ch * 8);
    *reinterpret_cast<uint2 *>(Vl + lrow * kSdVRow + ch * 8) = make_uint2(v.x, v.y);          // (136-byte rows: 8-byte aligned pieces)
    *reinterpret_cast<uint2 *>(Vl + lrow * kSdVRow + ch * 8 + 4) = make_uint2(v.z, v.w);
  }
  pl[tid] = obias[tid];
  pl[256 + tid] = gamma[tid];
  pl[512 + tid] = beta[tid];
  __syncthreads();
  const u16 *qb = q + (size_t)b * q_bstride, *rb = resid + (size_t)b * r_bstride;
  u16 *ob = out + (size_t)b * N * kSdC;
  // this workgroup's tokens: gridDim.x workgroups share the N tokens of the prompt in 16-token strips
  const int nstrip = N / 16;
  // The strip loop is software-pipelined (round 4): the q (and positional) fragments of the NEXT strip and the residual rows of THIS
  // strip are requested before this strip's matrix work, so neither global-load latency is exposed (as written first, every strip
  // started with a load -> MFMA dependency and its epilogue with another: 1.2 - 1.9 ms per 1024 prompts at 2 waves per SIMD, bound
  // by neither HBM nor the matrix pipe).
  union QF { uint4 u; sd_bf16x8 v; u16 h[8]; };
  QF qn[KSN], pn[KSN];
  auto fetch_q = [&](int strip) __attribute__((always_inline)) {
    const int tok = strip * 16 + c;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      qn[ks].u = *reinterpret_cast<const uint4 *>(qb + (size_t)tok * q_ld + ks * 32 + g * 8);
      if (q_add) pn[ks].u = *reinterpret_cast<const uint4 *>(q_add + (size_t)tok * KD + ks * 32 + g * 8);
    }
  };
  const int strip0 = blockIdx.x * 4 + wave, sstep = gridDim.x * 4;
  if (strip0 < nstrip) fetch_q(strip0);
  for (int strip = strip0; strip < nstrip; strip += sstep) {
    const int tok = strip * 16 + c;
    // this strip's B fragments: q (+ W_q pe shared by every prompt; RAW: + pe itself), rounded to bf16
    QF qa[KSN];
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      qa[ks] = qn[ks];
      if (q_add) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qa[ks].h[e] = sd_f2bf(sd_bf2f(qa[ks].h[e]) + sd_bf2f(pn[ks].h[e]));
      }
    }
    union RR { uint4 u; u16 h[8]; };
    RR rr[8];                                                        // residual rows of this strip (tile pair p: channels 32 p + 8 g .. + 8), first used in the epilogue
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      if (S6D_SD_CHANPERM) {
        rr[pp].u = *reinterpret_cast<const uint4 *>(rb + (size_t)tok * kSdC + pp * 32 + g * 8);
      } else {                                                       // tiles 2 pp, 2 pp + 1: channels 16 nt + 4 g .. + 4, 8 bytes each
        const uint2 a = *reinterpret_cast<const uint2 *>(rb + (size_t)tok * kSdC + (2 * pp) * 16 + g * 4);
        const uint2 b2 = *reinterpret_cast<const uint2 *>(rb + (size_t)tok * kSdC + (2 * pp + 1) * 16 + g * 4);
        rr[pp].u = make_uint4(a.x, a.y, b2.x, b2.y);
      }
    }
    // ---- scores^T (64 x 16) = Kexp (64 x KD) . Q^T ------------------------------------------------------------
    sd_f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = sd_f32x4{0.f, 0.f, 0.f, 0.f};
      if (RAW) s[t] = *reinterpret_cast<const sd_f32x4 *>(cbias + (size_t)b * kSdJ + t * 16 + g * 4);   // rows j = t*16 + g*4 + r
    }
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const sd_bf16x8 ka = *reinterpret_cast<const sd_bf16x8 *>(Kl + (t * 16 + c) * KROWL + ((ks * 4 + g) ^ sd_kswz(c)) * 8);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qa[ks].v, s[t], 0, 0, 0);
      }
    }
    // the next strip's fragments fly under the softmax, the value product and the epilogue (requested here, where qa is dead: the
    // RAW form holds 2 x 8 fragments and would otherwise run out of registers)
    __builtin_amdgcn_sched_barrier(0);
    if (strip + sstep < nstrip) fetch_q(strip + sstep);
    // ---- softmax over the 8 token slots of a head: row j = tile*16 + g*4 + r = h*8 + slot -> h = 2*tile + (g>>1),
    //      slot = (g&1)*4 + r: four registers here and four in the lane 16 away --------------------------------------
    union { sd_bf16x8 v; u16 h[8]; } pb[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = ((g & 1) * 4 + r < n_tok) ? s[t][r] : -1e30f;
      float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
      m = fmaxf(m, __shfl_xor(m, 16));
      float e[4], sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[r] = __expf(v[r] - m);
        sum += e[r];
      }
      sum += __shfl_xor(sum, 16);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int r = 0; r < 4; ++r) pb[t >> 1].h[(t & 1) * 4 + r] = sd_f2bf(e[r] * inv);
    }
    // ---- O^T (256 x 16) = V'^T (256 x 64) . P^T: k-step ks covers j in [32ks, 32ks+32); element e < 4 is
    //      j = 32ks + g*4 + e, e >= 4 is j = 32ks + 16 + g*4 + (e-4): exactly what pb[ks] already holds ---------------
    sd_f32x4 o[16];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
      o[nt] = sd_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        union { sd_bf16x8 v; uint2 d[2]; } va;
        const u16 *vr = Vl + ((S6D_SD_LDSFIX || !S6D_SD_CHANPERM) ? nt * 16 + c : chan(nt, c)) * kSdVRow + ks * 32 + g * 4;      // LDS row nt * 16 + c holds channel chan(nt, c)
        va.d[0] = *reinterpret_cast<const uint2 *>(vr);
        va.d[1] = *reinterpret_cast<const uint2 *>(vr + 16);
        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va.v, pb[ks].v, o[nt], 0, 0, 0);
      }
      if ((nt & 3) == 3) __builtin_amdgcn_sched_barrier(0);         // keep the fragment reads of later tiles from piling up
    }
    // ---- + out_proj bias + residual, LayerNorm over the 256 channels of token c (channel n = nt*16 + g*4 + r:
    //      64 here, the rest in the lanes 16 / 32 / 48 away) -----------------------------------------------------------
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[nt][r] += pl[chan(nt, g * 4 + r)] + sd_bf2f(rr[nt >> 1].h[(nt & 1) * 4 + r]);
        sum += o[nt][r];
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / kSdC);
    float var = 0.f;
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = o[nt][r] - mean;
        var += d * d;
      }
    var += __shfl_xor(var, 16);
    var += __shfl_xor(var, 32);
    const float rstd = rsqrtf(var * (1.0f / kSdC) + eps);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      union { uint4 u; u16 h[8]; } w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int nt = 2 * pp + (e >> 2), r = e & 3;
        const int n = chan(nt, 4 * g + r);                           // (permuted: = pp * 32 + g * 8 + e)
        w.h[e] = sd_f2bf((o[nt][r] - mean) * rstd * pl[256 + n] + pl[512 + n]);
      }
      if (S6D_SD_CHANPERM) {
        *reinterpret_cast<uint4 *>(ob + (size_t)tok * kSdC + pp * 32 + g * 8) = w.u;
      } else {
        *reinterpret_cast<uint2 *>(ob + (size_t)tok * kSdC + (2 * pp) * 16 + g * 4) = make_uint2(w.u.x, w.u.y);
        *reinterpret_cast<uint2 *>(ob + (size_t)tok * kSdC + (2 * pp + 1) * 16 + g * 4) = make_uint2(w.u.z, w.u.w);
      }
    }
  }
}

// erf-form GELU: gelu_erf of csrc/s6d_common.h (relative error 6e-6, far below the bf16 grid of the inputs)

constexpr int kUpC1 = 64, kUpC2 = 32;
constexpr int kUpWRow = kUpC1 + 8;   // LDS row stride of W2^T (bf16): 144 B

// y0 (B,N,4*64) bf16, row stride y_ld: first transposed conv as a GEMM, columns ordered (dy, dx, c);  ln_w, ln_b (64) f32;
// w2t (128,64) bf16: row (dy2*2+dx2)*32 + ch of the second transposed conv, transposed; b2 (32) f32;
// hyper (B,M,32) f32 -> masks (B,M,4h,4w) f32.
__global__ __launch_bounds__(256, 2) void upscale_heads_kernel(const u16 *__restrict__ y0, const float *__restrict__ ln_w,
                                                            const float *__restrict__ ln_b, float ln_eps,
                                                            const u16 *__restrict__ w2t, const float *__restrict__ b2,
                                                            const float *__restrict__ hyper, int M, int h, int w,
                                                            int y_ld, float *__restrict__ masks) {
  __shared__ __attribute__((aligned(16))) u16 Wl[4 * kUpC2 * kUpWRow];
  __shared__ float hy[4 * kUpC2], lw[kUpC1], lb[kUpC1], bb[kUpC2];
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int N = h * w;
  for (int i = tid; i < 4 * kUpC2 * kUpC1 / 8; i += 256) {
    const int row = i / (kUpC1 / 8), ch = i - row * (kUpC1 / 8);
    *reinterpret_cast<uint4 *>(Wl + row * kUpWRow + (ch ^ sd_kswz(row)) * 8) = *reinterpret_cast<const uint4 *>(w2t + row * kUpC1 + ch * 8);
  }
  if (tid < 4 * kUpC2) hy[tid] = (tid < M * kUpC2) ? hyper[(size_t)b * M * kUpC2 + tid] : 0.f;
  if (tid < kUpC1) { lw[tid] = ln_w[tid]; lb[tid] = ln_b[tid]; }
  if (tid < kUpC2) bb[tid] = b2[tid];
  __syncthreads();
  // The hypernetwork product  logit[mask][pixel] = sum_ch hyper[mask][ch] v[ch][pixel]  runs on the matrix core as well: A = hyper
  // (rows = masks, bf16 hi + lo parts: two instructions keep the fp32 factor exact to 2^-17), B = the GELU output of the second
  // transposed conv, rounded to bf16 like every other activation of this path.  The accumulators of that conv hold, per lane
  // (token c, group g), channels half*16 + g*4 + r: the contraction index is taken in exactly that order, k' = g*8 + half*4 + r,
  // so the B fragment is the lane's own eight values and no exchange is needed; mask m sits in A row 4 m, i.e. it comes out in
  // accumulator register 0 of lane group g = m -- the lanes that store mask g.
  union { sd_bf16x8 v; u16 hh[8]; } hya_hi, hya_lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = (e >> 2) * 16 + g * 4 + (e & 3);
    const float hv = ((c & 3) == 0 && (c >> 2) < M) ? hy[(c >> 2) * kUpC2 + ch] : 0.f;
    const u16 hi = sd_f2bf(hv);
    hya_hi.hh[e] = hi;
    hya_lo.hh[e] = sd_f2bf(hv - sd_bf2f(hi));
  }
  // a strip = 16 tokens x one sub-pixel (dy, dx): 4N/16 strips per prompt.  (Round 6, measured and not kept: a wave taking the four
  // strips of a token group and writing the group's 4 x 4 logits as 16-byte stores, 256 contiguous bytes per mask row and
  // instruction instead of 8-byte pieces: 1732 -> 1764 us per 1024 prompts -- the kernel is bound by its 3.2 G exact-erf GELU
  // evaluations, not by the store granularity, unlike img2tok_kernel.)
  const int nstrip = N / 16 * 4;
  // software-pipelined like img2tok_kernel (round 4): the two 16-byte pieces of the NEXT strip are requested before this strip's
  // arithmetic (every strip used to start with a load -> LayerNorm dependency)
  union YA { uint4 u; u16 hh[8]; };
  YA yn[2];
  auto fetch_y = [&](int strip) __attribute__((always_inline)) {
    const int sp = strip & 3, tok = (strip >> 2) * 16 + c;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      yn[ks].u = *reinterpret_cast<const uint4 *>(y0 + ((size_t)b * N + tok) * y_ld + sp * kUpC1 + ks * 32 + g * 8);
  };
  const int strip0 = blockIdx.x * 4 + wave, sstep = gridDim.x * 4;
  if (strip0 < nstrip) fetch_y(strip0);
  for (int strip = strip0; strip < nstrip; strip += sstep) {
    const int sp = strip & 3, tok = (strip >> 2) * 16 + c;           // token of this lane's column
    // ---- LayerNorm2d + GELU on the 64-vector of (token, sub-pixel): 16 channels here, 48 in the lanes 16/32/48 away
    float x[2][8];
    float sum = 0.f;
    YA ya[2] = {yn[0], yn[1]};
    if (strip + sstep < nstrip) fetch_y(strip + sstep);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const YA a = ya[ks];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[ks][e] = sd_bf2f(a.hh[e]);
        sum += x[ks][e];
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / kUpC1);
    float var = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = x[ks][e] - mean;
        var += d * d;
      }
    var += __shfl_xor(var, 16);
    var += __shfl_xor(var, 32);
    const float rstd = __frsqrt_rn(var * (1.0f / kUpC1) + ln_eps);
    union { sd_bf16x8 v; u16 hh[8]; } ua[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        float gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ch = ks * 32 + g * 8 + e + i;
          gv[i] = (x[ks][e + i] - mean) * rstd * lw[ch] + lb[ch];
        }
        gelu_erf4(gv);                                               // four evaluations on packed fp32 instructions, same bits
#pragma unroll
        for (int i = 0; i < 4; ++i) ua[ks].hh[e + i] = sd_f2bf(gv[i]);
      }
    // ---- second transposed conv: V^T (128 x 16) = W2^T (128 x 64) . U^T; row n = s2*32 + ch, s2 = dy2*2 + dx2 -----
    float mine[4];                                                   // logit of mask g at the four sub-pixels s2 of this token
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      union { sd_bf16x8 v; u16 hh[8]; } vb;                          // GELU(conv) of this lane's 8 channels, k' order
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int nt = s2 * 2 + half;
        sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const sd_bf16x8 wa = *reinterpret_cast<const sd_bf16x8 *>(Wl + (nt * 16 + c) * kUpWRow + ((ks * 4 + g) ^ sd_kswz(c)) * 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, ua[ks].v, acc, 0, 0, 0);
        }
        // C layout: row n_local = g*4 + r -> ch = half*16 + g*4 + r, col = token c
        float gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gv[r] = acc[r] + bb[half * 16 + g * 4 + r];
        gelu_erf4(gv);
#pragma unroll
        for (int r = 0; r < 4; ++r) vb.hh[half * 4 + r] = sd_f2bf(gv[r]);
      }
      sd_f32x4 lg = {0.f, 0.f, 0.f, 0.f};
      lg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hya_hi.v, vb.v, lg, 0, 0, 0);
      lg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hya_lo.v, vb.v, lg, 0, 0, 0);
      mine[s2] = lg[0];                                              // row 4 g = mask g, column = token c
    }
    if (g < M) {
      const int ty = tok / w, tx = tok - ty * w;
      const int py = 4 * ty + 2 * (sp >> 1), px = 4 * tx + 2 * (sp & 1);
      float *dst = masks + (((size_t)b * M + g) * (4 * h) + py) * (size_t)(4 * w) + px;
      *reinterpret_cast<float2 *>(dst) = make_float2(mine[0], mine[1]);
      *reinterpret_cast<float2 *>(dst + 4 * w) = make_float2(mine[2], mine[3]);
    }
  }
}

// token -> image attention of the two-way block, before out_proj: 8 heads x (<= 8 prompt tokens) = 64 (head, token)
// pairs = the 64 lanes of a wave.  Lane (h, t) keeps q_t's 16 head-h channels in registers and walks the image tokens:
// score, online softmax and the 16-channel value accumulation are all lane-local.  The 8 lanes of a head need the same
// 32 bytes of k and v, so each wave stages 8-token tiles through its own LDS slice (every lane fetches one distinct
// 16-byte piece; k + pe is rounded once there, not once per lane) and the next tile's pieces are in flight during the
// arithmetic.  16 waves split the N tokens of a prompt; their (max, sum, acc) partials meet in LDS.
// qt (B,8,128) f32 projected prompt tokens (rows >= T are ignored by the caller); k, v: bf16 (Bk,N,ld) slices at
// k_off / v_off of a fused projection (+ k_pe (N,128) bf16 or null); out (B,8,128) f32.
constexpr int kT2IWaves = 16;
constexpr int kT2ITile = 8;                       // tokens per staged tile: 8 x (256 B k + 256 B v) = 4 KB per wave
constexpr int kT2IRow = 2 * kSdD + 8;             // LDS row (bf16): [k | v] + 16 B pad
__global__ __launch_bounds__(kT2IWaves * 64) void tok2img_kernel(const float *__restrict__ qt, const u16 *__restrict__ kv,
                                                                int ld, int k_off, int v_off, long kv_bstride,
                                                                const u16 *__restrict__ k_pe, int N, float scale,
                                                                float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char t2i_smem[];
  float (*part)[64][18] = reinterpret_cast<float (*)[64][18]>(t2i_smem);   // [wave][lane][m, l, acc[16]]: 72 KB
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  u16 *tile = reinterpret_cast<u16 *>(t2i_smem + (size_t)kT2IWaves * 64 * 18 * 4) + (size_t)wave * kT2ITile * kT2IRow;
  const int h = lane >> 3, t = lane & 7;
  float q[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) q[d] = qt[((size_t)b * 8 + t) * kSdD + h * 16 + d] * scale;
  const u16 *base = kv + (size_t)b * kv_bstride;
  float m = -1e30f, l = 0.f, acc[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) acc[d] = 0.f;
  const int per = ((N + kT2IWaves - 1) / kT2IWaves + kT2ITile - 1) / kT2ITile * kT2ITile;
  const int n0 = wave * per, n1 = min(n0 + per, N);
  // staging: 8 tokens x 32 pieces of 16 B (16 of k, 16 of v) = 256 pieces, 4 per lane: piece i -> token i>>5, slot i&31
  uint4 pk[4];
  auto fetch = [&](int nt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = lane + j * 64, tk = i >> 5, sl = i & 31;
      const int nc = min(nt + tk, N - 1);
      const u16 *row = base + (size_t)nc * ld;
      pk[j] = *reinterpret_cast<const uint4 *>(row + (sl < 16 ? k_off + sl * 8 : v_off + (sl - 16) * 8));
    }
    if (k_pe) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = lane + j * 64, tk = i >> 5, sl = i & 31;
        if (sl < 16) {                                                // k pieces only
          const int nc = min(nt + tk, N - 1);
          const uint4 e = *reinterpret_cast<const uint4 *>(k_pe + (size_t)nc * kSdD + sl * 8);
          union { uint4 u; u16 hh[8]; } a, c2;
          a.u = pk[j];
          c2.u = e;
#pragma unroll
          for (int x = 0; x < 8; ++x) a.hh[x] = sd_f2bf(sd_bf2f(a.hh[x]) + sd_bf2f(c2.hh[x]));
          pk[j] = a.u;
        }
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = lane + j * 64, tk = i >> 5, sl = i & 31;
      *reinterpret_cast<uint4 *>(tile + tk * kT2IRow + sl * 8) = pk[j];
    }
  };
  if (n0 < n1) fetch(n0);
  for (int nt = n0; nt < n1; nt += kT2ITile) {
    commit();
    // hipemu: wave rendezvous (the wave's own LDS tile: its stores and loads execute in program order on the GPU)
    if (nt + kT2ITile < n1) fetch(nt + kT2ITile);
    const int cnt = min(kT2ITile, n1 - nt);
    for (int k = 0; k < cnt; ++k) {
      union { uint4 u[2]; u16 hh[16]; } kk, vv;
      const u16 *row = tile + k * kT2IRow;
      kk.u[0] = *reinterpret_cast<const uint4 *>(row + h * 16);
      kk.u[1] = *reinterpret_cast<const uint4 *>(row + h * 16 + 8);
      vv.u[0] = *reinterpret_cast<const uint4 *>(row + kSdD + h * 16);
      vv.u[1] = *reinterpret_cast<const uint4 *>(row + kSdD + h * 16 + 8);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int d = 0; d < 16; d += 2) {
        s0 += q[d] * sd_bf2f(kk.hh[d]);
        s1 += q[d + 1] * sd_bf2f(kk.hh[d + 1]);
      }
      const float s = s0 + s1;
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn), p = __expf(s - mn);
      l = l * alpha + p;
#pragma unroll
      for (int d = 0; d < 16; ++d) acc[d] = acc[d] * alpha + p * sd_bf2f(vv.hh[d]);
      m = mn;
    }
    // hipemu: wave rendezvous (the wave's own LDS tile: its stores and loads execute in program order on the GPU)
  }
  part[wave][lane][0] = m;
  part[wave][lane][1] = l;
#pragma unroll
  for (int d = 0; d < 16; ++d) part[wave][lane][2 + d] = acc[d];
  __syncthreads();
  if (wave == 0) {
    float M = -1e30f;
    for (int w = 0; w < kT2IWaves; ++w) M = fmaxf(M, part[w][lane][0]);
    float L = 0.f, o[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] = 0.f;
    for (int w = 0; w < kT2IWaves; ++w) {
      const float f = __expf(part[w][lane][0] - M);
      L += part[w][lane][1] * f;
#pragma unroll
      for (int d = 0; d < 16; ++d) o[d] += part[w][lane][2 + d] * f;
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int d = 0; d < 16; ++d) out[((size_t)b * 8 + t) * kSdD + h * 16 + d] = o[d] * inv;
  }
}

// ---- token -> image attention on the RAW image tokens (round 4) -------------------------------------------------------------------
// Reference: Attention.forward of segment_anything/modeling/transformer.py:185-232 as used by TwoWayAttentionBlock step (2) and the
// transformer's final attention: softmax(q_t . k_n / 4) v_n with k_n = W_k (x_n + pe_n) + b_k, v_n = W_v x_n + b_v per prompt.
// tok2img_kernel above reads k and v that a GEMM over ALL B x 4096 image tokens wrote (2.1 GB read + 2.1 GB written per use).
// The projections fold into the 8 x 8 (head, token) queries instead:
//     q_t,h . k_n = (W_k,h^T q_t,h) . (x_n + pe_n) + const      (the b_k term is the same for every n: softmax drops it)
//     sum_n p_n v_n = W_v,h (sum_n p_n x_n) + b_v,h              (sum_n p_n = 1)
// so the kernel attends 64 folded 256-d queries q'_j (j = head * 8 + slot, pre-multiplied by scale * log2 e, zero rows for unused
// slots) against the raw tokens and returns y_j = sum_n p_jn x_n: one read of x (2.1 GB per 1024 prompts), no k / v tensor; the
// caller applies W_v per head to the 64 x 256 result (a 1024 x 64 x 256 x 16 product).  Matrix work: 4 x the scalar kernel's
// flops, on the matrix cores: S^T = (X + PE) Q'^T and O^T += X^T P^T per 64-token tile, a wave per 16 query slots (2 heads), the
// tile shared by the four waves through LDS (one image of x + pe for the K fragments, one of x for the transposed V reads).
constexpr int kT2RTile = 64;                      // image tokens per tile
constexpr int kT2RKRow = kSdC + 8;                // x + pe image row (bf16): 528 B, conflict-free ds_read_b128 fragments
constexpr int kT2RVRow = kSdC + 16;               // x image row: 17 x 32 B, an odd multiple of 32 B for ds_read_b64_tr_b16
constexpr int kT2RLds = kT2RTile * (kT2RKRow + kT2RVRow) * 2;
typedef __attribute__((ext_vector_type(4))) short sd_s16x4;
#define S6D_SD_LDS(T) __attribute__((address_space(3))) T

// qp (B,64,256) bf16; x (Bx,N,256) bf16 rows of stride ld, Bx in {1, B} (x_bstride = 0: shared); pe (N,256) bf16 or null;
// y (B,64,256) f32.  N % 64 == 0.  One 4-wave workgroup per prompt.
__global__ __launch_bounds__(256, 2) void tok2img_raw_kernel(const u16 *__restrict__ qp, const u16 *__restrict__ x, int ld,
                                                            long x_bstride, const u16 *__restrict__ pe, int N,
                                                            float *__restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char t2r_smem[];
  u16 *Kl = reinterpret_cast<u16 *>(t2r_smem);                      // [64][kT2RKRow]  x + pe
  u16 *Vl = Kl + kT2RTile * kT2RKRow;                               // [64][kT2RVRow]  x
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  // B fragments of this wave's 16 query slots: row c, k elements g*8 .. g*8+7 of each of the 8 k-steps
  sd_bf16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    union { uint4 u; sd_bf16x8 v; } t;
    t.u = *reinterpret_cast<const uint4 *>(qp + ((size_t)b * kSdJ + wave * 16 + c) * kSdC + ks * 32 + g * 8);
    qf[ks] = t.v;
  }
  const u16 *xb = x + (size_t)b * x_bstride;
  // staging: 64 tokens x 32 chunks of 16 B = 2048 chunks, 8 per thread: chunk i = tid + 256 j -> token i >> 5, chunk i & 31
  uint4 px[8], pp[8];
  auto fetch = [&](int nt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = tid + 256 * j, tk = i >> 5, ch = i & 31;
      px[j] = *reinterpret_cast<const uint4 *>(xb + (size_t)(nt + tk) * ld + ch * 8);
      if (pe) pp[j] = *reinterpret_cast<const uint4 *>(pe + (size_t)(nt + tk) * kSdC + ch * 8);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = tid + 256 * j, tk = i >> 5, ch = i & 31;
      *reinterpret_cast<uint4 *>(Vl + tk * kT2RVRow + ch * 8) = px[j];
      union { uint4 u; u16 h[8]; } a, e;
      a.u = px[j];
      if (pe) {
        e.u = pp[j];
#pragma unroll
        for (int k = 0; k < 8; ++k) a.h[k] = sd_f2bf(sd_bf2f(a.h[k]) + sd_bf2f(e.h[k]));
      }
      *reinterpret_cast<uint4 *>(Kl + tk * kT2RKRow + (ch ^ sd_kswz(tk)) * 8) = a.u;
    }
  };
  float m_run = -1e30f;
  sd_f32x4 lacc = {0.f, 0.f, 0.f, 0.f}, oacc[16];
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) oacc[dt] = sd_f32x4{0.f, 0.f, 0.f, 0.f};
  union { sd_bf16x8 v; u16 h[8]; } ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones.h[i] = 0x3f80;                  // bf16 1.0
  fetch(0);
  for (int nt = 0; nt < N; nt += kT2RTile) {
    commit();
    __syncthreads();                                                // the tile is in LDS
    if (nt + kT2RTile < N) fetch(nt + kT2RTile);                    // the next one flies under this one's arithmetic
    // ---- S^T (64 tokens x 16 slots) = (X + PE) Q'^T: acc[r] = score of token sub*16 + g*4 + r against slot c ----------------
    float s[4][4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const sd_bf16x8 ka = *reinterpret_cast<const sd_bf16x8 *>(Kl + (sub * 16 + c) * kT2RKRow + ((ks * 4 + g) ^ sd_kswz(c)) * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) s[sub][r] = acc[r];
    }
    // ---- online softmax over the tokens (queries were pre-multiplied by scale * log2 e); the rescale of the accumulators is skipped
    //      while no slot's maximum grew by more than 2^6 (P stays <= 64: exact in bf16's exponent range) --------------------------
    float mx = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[sub][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__any(mx - m_run > 6.0f)) {                                 // wave-uniform
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      lacc *= alpha;
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) oacc[dt] *= alpha;
      m_run = m_new;
    }
    union { sd_bf16x8 v; u16 h[8]; } pb[2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) pb[sub >> 1].h[(sub & 1) * 4 + r] = sd_f2bf(__builtin_amdgcn_exp2f(s[sub][r] - m_run));
    // ---- O^T (256 x 16 slots) += X^T P^T over two 32-token steps; the row sums ride an all-ones A fragment ------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u16 *vrow = Vl + (32 * j + g * 4 + (c >> 2)) * kT2RVRow + (c & 3) * 4;
      lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, pb[j].v, lacc, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) {
        union { sd_bf16x8 v; sd_s16x4 q[2]; } va;
        va.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_SD_LDS(sd_s16x4) *)(vrow + dt * 16));
        va.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_SD_LDS(sd_s16x4) *)(vrow + 16 * kT2RVRow + dt * 16));
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va.v, pb[j].v, oacc[dt], 0, 0, 0);
      }
    }
    __syncthreads();                                                // every wave is done with the tile: it may be overwritten
  }
  // y[b][slot = wave*16 + c][d = dt*16 + g*4 + r]
  const float inv = 1.0f / lacc[0];
  float *yo = y + ((size_t)b * kSdJ + wave * 16 + c) * kSdC + g * 4;
#pragma unroll
  for (int dt = 0; dt < 16; ++dt)
    *reinterpret_cast<float4 *>(yo + dt * 16) = make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
}

// ---- mask post-processing of the automatic mask generator, fused ---------------------------------------------------
// Reference: Sam.postprocess_masks (modeling/sam.py:133-162: bilinear to the padded square, crop, bilinear to the frame),
// then on the full-resolution logits calculate_stability_score, `> mask_threshold` and batched_mask_to_box
// (utils/amg.py:156-176, 303-346; automatic_mask_generator.py:281-312).  As written that is 805 MB of fp32 per batch of
// 64 prompts.  Here every frame pixel is evaluated straight from the 256 x 256 logits (4 taps of the intermediate image,
// each 4 taps of the logits) and only the binary mask, two counts and a bounding box per mask leave the kernel.
// The arithmetic is ATen's CPU upsample_bilinear2d form, reproduced operation for operation
//     src = scale * (dst + 0.5) - 0.5 (clamped at 0), w1 = src - floor(src), w0 = 1 - w1
//     v = fma(wy0, fma(wx0, v00, wx1 * v01), wy1 * fma(wx0, v10, wx1 * v11))
// so the thresholded masks are bit-identical to the oracle's.
struct BilTap { int i0, i1; float w0, w1; };
__device__ __forceinline__ BilTap bil_tap(int dst, float scale, int in_n) {
  float sidx = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
  sidx = sidx < 0.f ? 0.f : sidx;
  BilTap t;
  t.i0 = (int)sidx;
  t.i1 = t.i0 + (t.i0 < in_n - 1 ? 1 : 0);
  t.w1 = __fsub_rn(sidx, (float)t.i0);
  t.w0 = __fsub_rn(1.0f, t.w1);
  return t;
}
__device__ __forceinline__ float bil_mix(float w0, float a, float w1, float b) { return __fmaf_rn(w0, a, __fmul_rn(w1, b)); }

// Round 4: a lane owns a frame COLUMN and walks a strip of kPostRows frame rows.  Everything that depends on the column only (the
// horizontal taps of both stages) is computed once per strip; everything that depends on the row only (the vertical taps of both
// stages) comes from two small LDS tables built once per workgroup and is wave-uniform (readfirstlane: the cache logic below
// runs on the scalar unit).  The horizontally mixed logits of the current pair of low-resolution rows and the values of the last
// two rows of the intermediate image are kept in registers and re-used while the walk stays on them (a low-resolution row pair
// serves ~4 intermediate rows ~2.5 frame rows): 4-6 loads and ~40 VALU instructions per pixel instead of 16 and ~150 (the
// round-3 kernel recomputed all 16 taps and every index per pixel and was bound by the vector issue rate: 2.8 ms per 3072
// masks).  Every value is produced by the same operations in the same order as before, so the masks keep the oracle's bits.
// (Tried at the end of round 3: one wave per frame row with the row's taps computed once, four consecutive pixels per lane and
// their mask bytes stored as one word -- bit-identical, and 2.7x SLOWER, 7.6 against 2.8 ms per 3072 masks: with a lane per
// pixel the taps of neighbouring lanes fall into the same few cache lines, with a lane per quad they do not.)
constexpr int kPostRows = 32;      // frame rows per workgroup
constexpr int kPostThreads = 320;  // 5 waves: 640 columns in two passes
constexpr int kPostMaxInter = 2 * kPostRows + 2;   // intermediate rows a strip may touch: the entry point shortens the strip for frames
                                                   // more than twice as small as the resized input (rows * in_h / H + 2 <= this)

__global__ void mask_post_init_kernel(int *stats, int Bm, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Bm) {
    int *s = stats + (size_t)i * 6;
    s[0] = 0; s[1] = 0; s[2] = W; s[3] = H; s[4] = -1; s[5] = -1;
  }
}

// Mask m = b * ch_count + c reads the logits plane (b * ch_total + ch_first + c): the multimask slice masks[:, 1:] of the
// decoder's (B, 4, n, n) output is read in place (the contiguous copy of the slice was 0.57 ms per 1024 prompts).
__global__ __launch_bounds__(kPostThreads) void mask_post_kernel(const float *__restrict__ low, int ch_total, int ch_first,
                                                                 int ch_count, int n, int img, int ih, int iw,
                                                                 int H, int W, int strip, float thr, float off,
                                                                 unsigned char *__restrict__ masks, int *__restrict__ stats) {
  __shared__ int red[6][kPostThreads / 64];
  __shared__ BilTap tabB[kPostRows];          // frame row -> rows of the intermediate image
  __shared__ BilTap tabA[kPostMaxInter];      // intermediate row -> rows of the logits
  const int m = blockIdx.y, y_base = blockIdx.x * strip, tid = threadIdx.x;
  const int rows = min(strip, H - y_base);
  const float *L = low + ((size_t)(m / ch_count) * ch_total + ch_first + m % ch_count) * n * n;
  const float sA = (float)n / (float)img, sBy = (float)ih / (float)H, sBx = (float)iw / (float)W;
  const int Y_first = bil_tap(y_base, sBy, ih).i0;
  if (tid < rows) tabB[tid] = bil_tap(y_base + tid, sBy, ih);
  if (tid >= 64 && tid < 64 + kPostMaxInter) {
    const int Y = min(Y_first + tid - 64, ih - 1);
    tabA[tid - 64] = bil_tap(Y, sA, n);
  }
  __syncthreads();
  int inter = 0, uni = 0, xmin = W, ymin = H, xmax = -1, ymax = -1;
  for (int x0 = 0; x0 < W; x0 += kPostThreads) {          // every lane of a wave walks the strip (the cache logic is wave-uniform);
    const bool live = x0 + tid < W;                       // lanes past the last column compute on it and write nothing
    const int x = live ? x0 + tid : W - 1;
    const BilTap bx = bil_tap(x, sBx, iw);
    const BilTap ax0 = bil_tap(bx.i0, sA, n), ax1 = bil_tap(bx.i1, sA, n);
    int cl0 = -1, cl1 = -1;                   // logits rows the h** registers hold (uniform)
    float h00 = 0.f, h01 = 0.f, h10 = 0.f, h11 = 0.f;     // h[row 0/1][column X0/X1]: horizontally mixed logits
    int cY0 = -1, cY1 = -1;                   // intermediate rows the v*c registers hold (uniform)
    float v0c = 0.f, v1c = 0.f;
    auto hrow = [&](int row, float &hx0, float &hx1) {
      const float *r = L + (size_t)row * n;
      hx0 = bil_mix(ax0.w0, r[ax0.i0], ax0.w1, r[ax0.i1]);
      hx1 = bil_mix(ax1.w0, r[ax1.i0], ax1.w1, r[ax1.i1]);
    };
    auto inter_row = [&](int Y) -> float {    // value of the intermediate image at (Y, this lane's frame column)
      const BilTap *ta = &tabA[Y - Y_first];
      const int i0 = __builtin_amdgcn_readfirstlane(ta->i0), i1 = __builtin_amdgcn_readfirstlane(ta->i1);
      const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ta->w0)));
      const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ta->w1)));
      if (i0 != cl0) {
        if (i0 == cl1 && cl1 != cl0) { h00 = h10; h01 = h11; }
        else hrow(i0, h00, h01);
        if (i1 == i0) { h10 = h00; h11 = h01; }
        else hrow(i1, h10, h11);
        cl0 = i0; cl1 = i1;
      }
      const float a0 = bil_mix(w0, h00, w1, h10);         // intermediate (Y, X_0)
      const float a1 = bil_mix(w0, h01, w1, h11);         // intermediate (Y, X_1)
      return bil_mix(bx.w0, a0, bx.w1, a1);
    };
    for (int r = 0; r < rows; ++r) {
      const int y = y_base + r;
      const int Y0 = __builtin_amdgcn_readfirstlane(tabB[r].i0), Y1 = __builtin_amdgcn_readfirstlane(tabB[r].i1);
      const float wy0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tabB[r].w0)));
      const float wy1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tabB[r].w1)));
      float v0, v1;
      if (Y0 == cY0) v0 = v0c;
      else if (Y0 == cY1) v0 = v1c;
      else v0 = inter_row(Y0);
      if (Y1 == Y0) v1 = v0;
      else if (Y1 == cY1) v1 = v1c;
      else v1 = inter_row(Y1);
      cY0 = Y0; v0c = v0; cY1 = Y1; v1c = v1;
      const float v = bil_mix(wy0, v0, wy1, v1);
      const bool on = live && v > thr;
      if (live) masks[((size_t)m * H + y) * W + x] = on ? 1 : 0;
      inter += live && v > thr + off;
      uni += live && v > thr - off;
      if (on) {
        xmin = min(xmin, x); xmax = max(xmax, x);
        ymin = min(ymin, y); ymax = max(ymax, y);
      }
    }
  }
  // workgroup fold, then one atomic per statistic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inter += __shfl_xor(inter, o);
    uni += __shfl_xor(uni, o);
    xmin = min(xmin, __shfl_xor(xmin, o));
    ymin = min(ymin, __shfl_xor(ymin, o));
    xmax = max(xmax, __shfl_xor(xmax, o));
    ymax = max(ymax, __shfl_xor(ymax, o));
  }
  const int wave = tid >> 6;
  if ((tid & 63) == 0) {
    red[0][wave] = inter; red[1][wave] = uni; red[2][wave] = xmin; red[3][wave] = ymin; red[4][wave] = xmax;
    red[5][wave] = ymax;
  }
  __syncthreads();
  if (tid == 0) {
    int *s = stats + (size_t)m * 6;
    int a = 0, b = 0, c = W, d = H, e = -1, f = -1;
    for (int w = 0; w < kPostThreads / 64; ++w) {
      a += red[0][w]; b += red[1][w];
      c = min(c, red[2][w]); d = min(d, red[3][w]); e = max(e, red[4][w]); f = max(f, red[5][w]);
    }
    atomicAdd(s + 0, a);
    atomicAdd(s + 1, b);
    atomicMin(s + 2, c);
    atomicMin(s + 3, d);
    atomicMax(s + 4, e);
    atomicMax(s + 5, f);
  }
}

// ---- box NMS (torchvision.ops.nms semantics; the generator's duplicate removal, automatic_mask_generator.py:251-257) ---
// Phase 1: for every pair i < j in score order, bit j of row i says IoU(i, j) > threshold (area = (x2-x1)(y2-y1), no +1,
// float32, as torchvision's devIoU).  Phase 2: one wave walks the rows in order; lane l owns the 64-bit word l of the
// "removed" set, so taking a box is one OR per lane.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, const long *__restrict__ order, int N,
                                                      float thresh, unsigned long long *__restrict__ mask) {
  const int i = blockIdx.x, wcol = blockIdx.y, lane = threadIdx.x;
  const int nw = (N + 63) / 64;
  const int j = wcol * 64 + lane;
  const float *a = boxes + order[i] * 4;
  bool hit = false;
  if (j < N && j > i) {
    const float *b = boxes + order[j] * 4;
    const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
    const float inter = w * h, sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
    hit = inter / (sa + sb - inter) > thresh;
  }
  const unsigned long long bits = __ballot(hit);
  if (lane == 0) mask[(size_t)i * nw + wcol] = bits;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long *__restrict__ mask, int N,
                                                      unsigned char *__restrict__ keep) {
  const int lane = threadIdx.x, nw = (N + 63) / 64;
  unsigned long long removed[4] = {0, 0, 0, 0};                     // words lane, lane+64, ... : N <= 16384
  for (int i = 0; i < N; ++i) {
    const int w = i >> 6;
    const unsigned long long word = __shfl(removed[w >> 6], w & 63);   // every lane learns bit i of the removed set
    const bool dead = (word >> (i & 63)) & 1ull;
    if (lane == 0) keep[i] = dead ? 0 : 1;
    if (!dead) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ww = lane + k * 64;
        if (ww < nw) removed[k] |= mask[(size_t)i * nw + ww];
      }
    }
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" long s6d_nms_workspace_bytes(int N) { return (long)N * ((N + 63) / 64) * 8; }

extern "C" int s6d_nms_f32(const float *boxes, const int64_t *order, int N, float iou_threshold, void *workspace,
                           unsigned char *keep_sorted, void *stream) {
  if (N < 0 || N > 16384) return N < 0 ? S6D_EINVAL : S6D_EUNSUPPORTED;
  if (N == 0) return S6D_OK;
  if (!boxes || !order || !workspace || !keep_sorted) return S6D_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(N, (N + 63) / 64), dim3(64), 0, st, boxes, (const long *)order, N, iou_threshold,
                     (unsigned long long *)workspace);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long *)workspace, N, keep_sorted);
  return launch_status();
}

extern "C" int s6d_sam_mask_post_sel_f32(const float *low_res, int B, int ch_total, int ch_first, int ch_count, int n, int img_size,
                                         int in_h, int in_w, int H, int W, float mask_threshold, float stability_offset,
                                         unsigned char *masks, int32_t *stats, void *stream) {
  if (B < 0 || ch_total <= 0 || ch_first < 0 || ch_count <= 0 || ch_first + ch_count > ch_total || n <= 0 || img_size <= 0 ||
      in_h <= 0 || in_w <= 0 || in_h > img_size || in_w > img_size || H <= 0 || W <= 0)
    return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!low_res || !masks || !stats) return S6D_EINVAL;
  if ((long)B * ch_count > 65535L * 32) return S6D_EINVAL;
  const int Bm = B * ch_count;
  // frame rows per workgroup: 32, fewer when the frame is more than twice as small as the resized input (the table of the
  // intermediate rows a strip touches has kPostMaxInter entries: rows * in_h / H + 2 of them are used)
  int strip = (int)(((long)(kPostMaxInter - 2) * H) / in_h);
  strip = strip < 1 ? 1 : (strip > kPostRows ? kPostRows : strip);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(mask_post_init_kernel, dim3((Bm + 255) / 256), dim3(256), 0, st, stats, Bm, H, W);
  for (int m0 = 0; m0 < Bm; m0 += 65535 / ch_count * ch_count) {      // gridDim.y <= 65535, slabs of whole prompts
    const int mb = min(Bm - m0, 65535 / ch_count * ch_count);
    hipLaunchKernelGGL(mask_post_kernel, dim3((H + strip - 1) / strip, mb), dim3(kPostThreads), 0, st,
                       low_res + (size_t)(m0 / ch_count) * ch_total * n * n, ch_total, ch_first, ch_count, n, img_size, in_h, in_w,
                       H, W, strip, mask_threshold, stability_offset, masks + (size_t)m0 * H * W, stats + (size_t)m0 * 6);
  }
  return launch_status();
}

extern "C" int s6d_sam_mask_post_f32(const float *low_res, int Bm, int n, int img_size, int in_h, int in_w, int H, int W,
                                     float mask_threshold, float stability_offset, unsigned char *masks, int32_t *stats,
                                     void *stream) {
  if (Bm < 0) return S6D_EINVAL;
  return s6d_sam_mask_post_sel_f32(low_res, Bm, 1, 0, 1, n, img_size, in_h, in_w, H, W, mask_threshold, stability_offset, masks,
                                   stats, stream);
}

extern "C" int s6d_samdec_tok2img_f32(const float *qt, const void *kv, int ld, int k_off, int v_off, int kv_shared,
                                      const void *k_pe, int B, int N, float scale, float *out, void *stream) {
  if (B < 0 || N <= 0 || ld < kSdD || (ld % 8) || (k_off % 8) || (v_off % 8) || k_off < 0 || v_off < 0 ||
      k_off + kSdD > ld || v_off + kSdD > ld)
    return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!qt || !kv || !out) return S6D_EINVAL;
  const size_t lds = (size_t)kT2IWaves * 64 * 18 * sizeof(float) + (size_t)kT2IWaves * kT2ITile * kT2IRow * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tok2img_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL(tok2img_kernel, dim3(B), dim3(kT2IWaves * 64), lds, as_stream(stream), qt, (const u16 *)kv, ld, k_off,
                     v_off, kv_shared ? 0L : (long)N * ld, (const u16 *)k_pe, N, scale, out);
  return launch_status();
}

extern "C" int s6d_samdec_tok2img_raw_bf16(const void *qp, const void *x, int ld, int x_shared, const void *pe, int B, int N,
                                           float *y, void *stream) {
  if (B < 0 || N <= 0 || (N % kT2RTile) != 0 || ld < kSdC || (ld % 8) != 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!qp || !x || !y || (((uintptr_t)qp | (uintptr_t)x | (uintptr_t)pe | (uintptr_t)y) & 15)) return S6D_EINVAL;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tok2img_raw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kT2RLds);
  hipLaunchKernelGGL(tok2img_raw_kernel, dim3(B), dim3(256), kT2RLds, as_stream(stream), (const u16 *)qp, (const u16 *)x, ld,
                     x_shared ? 0L : (long)N * ld, (const u16 *)pe, N, y);
  return launch_status();
}

static int img2tok_launch(bool raw, const void *q, const void *q_add, const void *kexp, const float *cbias, const void *vpt,
                          const void *resid, const float *out_bias, const float *ln_w, const float *ln_b, float ln_eps, int B, int N,
                          int n_tok, int q_ld, int q_shared, int resid_shared, void *out, void *stream) {
  const int kd = raw ? kSdC : kSdD;
  if (B < 0 || N <= 0 || (N % 16) != 0 || n_tok <= 0 || n_tok > 8 || q_ld < kd || (q_ld % 8) != 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!q || !kexp || !vpt || !resid || !out_bias || !ln_w || !ln_b || !out || (raw && !cbias)) return S6D_EINVAL;
  const size_t lds = (size_t)(kSdJ * (kd + 8) + kSdC * kSdVRow) * 2 + 3 * 256 * 4;
  const int nstrip = N / 16;
  const int gx = nstrip >= 64 ? 16 : (nstrip + 3) / 4;              // 16 workgroups x 4 waves x 4 strips at N = 4096
  if (raw) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&img2tok_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(img2tok_kernel<true>, dim3(gx, B), dim3(256), lds, as_stream(stream), (const u16 *)q, (const u16 *)q_add,
                       (const u16 *)kexp, cbias, (const u16 *)vpt, (const u16 *)resid, out_bias, ln_w, ln_b, ln_eps, N, n_tok, q_ld,
                       q_shared ? 0L : (long)N * q_ld, resid_shared ? 0L : (long)N * kSdC, (u16 *)out);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&img2tok_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(img2tok_kernel<false>, dim3(gx, B), dim3(256), lds, as_stream(stream), (const u16 *)q, (const u16 *)q_add,
                       (const u16 *)kexp, cbias, (const u16 *)vpt, (const u16 *)resid, out_bias, ln_w, ln_b, ln_eps, N, n_tok, q_ld,
                       q_shared ? 0L : (long)N * q_ld, resid_shared ? 0L : (long)N * kSdC, (u16 *)out);
  }
  return launch_status();
}

extern "C" int s6d_samdec_img2tok_bf16(const void *q, const void *q_add, const void *kexp, const void *vpt,
                                       const void *resid, const float *out_bias, const float *ln_w, const float *ln_b,
                                       float ln_eps, int B, int N, int n_tok, int q_ld, int q_shared, int resid_shared,
                                       void *out, void *stream) {
  return img2tok_launch(false, q, q_add, kexp, nullptr, vpt, resid, out_bias, ln_w, ln_b, ln_eps, B, N, n_tok, q_ld, q_shared,
                        resid_shared, out, stream);
}

extern "C" int s6d_samdec_img2tok_raw_bf16(const void *x, const void *pe, const void *kexp256, const float *cbias, const void *vpt,
                                           const void *resid, const float *out_bias, const float *ln_w, const float *ln_b,
                                           float ln_eps, int B, int N, int n_tok, int x_ld, int x_shared, int resid_shared,
                                           void *out, void *stream) {
  return img2tok_launch(true, x, pe, kexp256, cbias, vpt, resid, out_bias, ln_w, ln_b, ln_eps, B, N, n_tok, x_ld, x_shared,
                        resid_shared, out, stream);
}

extern "C" int s6d_samdec_upscale_heads_bf16(const void *y0, const float *ln_w, const float *ln_b, float ln_eps,
                                             const void *w2t, const float *b2, const float *hyper, int B, int M, int h,
                                             int w, int y_ld, float *masks, void *stream) {
  if (B < 0 || M <= 0 || M > 4 || h <= 0 || w <= 0 || ((h * w) % 16) != 0 || y_ld < 4 * kUpC1 || (y_ld % 8) != 0)
    return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!y0 || !ln_w || !ln_b || !w2t || !b2 || !hyper || !masks) return S6D_EINVAL;
  const int nstrip = h * w / 16 * 4;
  const int gx = nstrip >= 256 ? 64 : (nstrip + 3) / 4;
  hipLaunchKernelGGL(upscale_heads_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), (const u16 *)y0, ln_w, ln_b,
                     ln_eps, (const u16 *)w2t, b2, hyper, M, h, w, y_ld, masks);
  return launch_status();
}
