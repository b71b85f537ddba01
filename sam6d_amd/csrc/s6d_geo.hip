// Geometric structure embedding of the PEM point transformer, fused (gfx950).
//
// Reference: GeometricStructureEmbedding.forward, Pose_Estimation_Model/model/transformer.py:334-349
//   d_emb = proj_d(sinusoid(d_idx))            (B,N,N,256)
//   a_emb = proj_a(sinusoid(a_idx)).max(k)     (B,N,N,3,256) -> 119 MB / instance intermediate
//   out   = d_emb + a_emb                       20.3 GFLOP per cloud as two fp32 GEMMs on 155 k rows
// Here the four sinusoidal embeddings of a point pair are generated on the fly (never stored), multiplied
// by W_d / W_a on the bf16 matrix cores with a 3-term split (x = x_hi + x_lo in bf16;
// x_hi*w_hi + x_lo*w_hi + x_hi*w_lo, fp32 accumulate: ~2^-17 relative per product, i.e. fp32-class
// results at 5x the fp32-MFMA rate), max-reduced over the k angular neighbours in registers and written
// once.  Workgroup = 64 point pairs x 256 outputs, 8 waves: wave (mt, nh) owns pairs [16mt,16mt+16) and
// outputs [128nh, 128nh+128).  Per 32-channel k-step: all waves first build the shared operands in LDS
// (sincos fragments hi/lo, weight slices hi/lo), then run 96 MFMAs each.
#include "s6d_common.h"

namespace s6d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

__device__ __forceinline__ void split_bf16(float x, u16 &hi, u16 &lo) {
  union { __bf16 b; u16 u; } h, l;
  h.b = (__bf16)x;
  const float xh = __uint_as_float(((unsigned)h.u) << 16);
  l.b = (__bf16)(x - xh);
  hi = h.u;
  lo = l.u;
}

// sin / cos of the fp32 product x*w (the reference's argument, transformer.py:275): two-constant Cody-Waite reduction
// to [-pi, pi] (k = rint(a / 2pi); r = a - k*2pi_hi - k*2pi_lo, both by FMA), then the hardware sine / cosine, which take
// their argument in revolutions.  Arguments reach ~500 rad (the background point sits 100 units away, sigma_d = 0.2):
// at that size the reduction keeps |error| < 2e-6 where libm-style sincosf spends ~45 instructions on it and this ~8.
__device__ __forceinline__ void fast_sincos(float a, float &sn, float &cs) {
  const float k = rintf(a * 0.15915494309189535f);
  float r = __fmaf_rn(-k, 6.28318548202514648f, a);                // 2pi rounded to fp32
  r = __fmaf_rn(-k, -1.74845553e-07f, r);                         // 2pi - fp32(2pi)
  const float rev = r * 0.15915494309189535f;
  sn = __builtin_amdgcn_sinf(rev);
  cs = __builtin_amdgcn_cosf(rev);
}

constexpr int GEO_C = 256;          // hidden dim
constexpr int GEO_ROW = 40;         // LDS row stride (bf16): 32 channels + 8 pad (80 B = 5 chunks of 16 B)
// A ds_read_b128 lane group is {rows 0-3,12-15 reading chunk g} + {rows 4-11 reading chunk g + 1}: with any odd row stride these
// meet on 3 of 16 bank slots (2-way conflict on every fragment read: SQ_LDS_BANK_CONFLICT was a quarter of the kernel's cycles).
// Rows with (row >> 2 ^ row >> 3) & 1 therefore keep their chunks pairwise swapped (same fix as csrc/s6d_attn.hip S6D_GLB_KSWZ).
#ifndef S6D_GEO_KSWZ
#define S6D_GEO_KSWZ 1
#endif
__device__ __forceinline__ int geo_swz(int row) { return S6D_GEO_KSWZ ? (((row >> 2) ^ (row >> 3)) & 1) : 0; }
constexpr int GEO_PAIRS = 64;       // point pairs per workgroup
constexpr int GEO_THREADS = 512;

// LDS map (bf16 elements)
constexpr int OFF_WDH = 0;
constexpr int OFF_WDL = OFF_WDH + GEO_C * GEO_ROW;
constexpr int OFF_WAH = OFF_WDL + GEO_C * GEO_ROW;
constexpr int OFF_WAL = OFF_WAH + GEO_C * GEO_ROW;
constexpr int OFF_AH = OFF_WAL + GEO_C * GEO_ROW;                  // [4 emb][64 pairs][ROW]
constexpr int OFF_AL = OFF_AH + 4 * GEO_PAIRS * GEO_ROW;
constexpr int GEO_LDS_ELEMS = OFF_AL + 4 * GEO_PAIRS * GEO_ROW;    // 61440 elems = 120 KB

// HALF: the embedding is written in IEEE half (2 bytes per channel) instead of float32 -- its only reader, rpe_attention_kernel, is
// bound by streaming it (12 reads of 1.4 GB per 32 instances); the arithmetic up to the store is the same.
// PRE (round 5): the weights arrive as their bf16 hi / lo parts (s6d_linear_split_weight_f32, made once per weight version by the
// caller) -- Wd / Wa then point at [hi (C x C) | lo (C x C)] bf16.  Until then EVERY workgroup split the same two 256 x 256 fp32
// matrices again in every k-step: ~200 of a thread's ~400 vector instructions per k-step beside 96 matrix instructions per wave.
template <bool HALF, bool PRE>
__global__ __launch_bounds__(GEO_THREADS) void geo_embed_kernel(const float *__restrict__ idx4, long NP,
                                                               const float *__restrict__ Wd, const float *__restrict__ bd,
                                                               const float *__restrict__ Wa, const float *__restrict__ ba,
                                                               const float *__restrict__ div_term, void *__restrict__ outv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u16 *lds = reinterpret_cast<u16 *>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int mt = wave & 3, nh = wave >> 2;
  const long pair0 = (long)blockIdx.x * GEO_PAIRS;

  // ---- per-thread constants of the operand builders -------------------------------------------------
  // weights: each k-step slice is 256 rows x 32 fp32 per matrix = 2048 float4 per matrix; 512 threads x 4
  // phase-1 items: (pair p, embedding e, channel group q of 8) = 64*4*4 = 1024 -> 2 per thread
  float xval[2];
  int item_p[2], item_e[2], item_q[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int it = tid + n * GEO_THREADS;
    item_q[n] = it & 3;
    item_e[n] = (it >> 2) & 3;
    item_p[n] = it >> 4;
    const long pr = min(pair0 + item_p[n], NP - 1);
    xval[n] = idx4[pr * 4 + item_e[n]];
  }
  float4 wreg[8];                                                  // prefetched weight slice (4 of W_d, 4 of W_a); PRE: 8 x 16 bytes of bf16
  auto wload = [&](int ks) {
    if (PRE) {
      // part q = 0..3 (W_d hi, W_d lo, W_a hi, W_a lo): 256 rows x 4 chunks of 8 bf16 per k-step = 1024 chunks, two per thread
      const u16 *parts[4] = {reinterpret_cast<const u16 *>(Wd), reinterpret_cast<const u16 *>(Wd) + GEO_C * GEO_C,
                             reinterpret_cast<const u16 *>(Wa), reinterpret_cast<const u16 *>(Wa) + GEO_C * GEO_C};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int it = tid + n * GEO_THREADS;                    // 0..1023: row = it / 4, chunk = it % 4
          const uint4 v = *reinterpret_cast<const uint4 *>(parts[q] + (size_t)(it >> 2) * GEO_C + ks * 32 + (it & 3) * 8);
          wreg[q * 2 + n] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
      return;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int it = tid + n * GEO_THREADS;                        // 0..2047: row = it/8, float4 col = it%8
      const int row = it >> 3, col = it & 7;
      wreg[n] = *reinterpret_cast<const float4 *>(Wd + (size_t)row * GEO_C + ks * 32 + col * 4);
      wreg[4 + n] = *reinterpret_cast<const float4 *>(Wa + (size_t)row * GEO_C + ks * 32 + col * 4);
    }
  };
  auto wstore = [&]() {
    if (PRE) {
      const int offs[4] = {OFF_WDH, OFF_WDL, OFF_WAH, OFF_WAL};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int it = tid + n * GEO_THREADS;
          const int row = it >> 2, ch = it & 3;
          const float4 f = wreg[q * 2 + n];
          *reinterpret_cast<uint4 *>(lds + offs[q] + row * GEO_ROW + ((ch ^ geo_swz(row)) * 8)) =
              make_uint4(__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w));
        }
      return;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int it = tid + n * GEO_THREADS;
      const int row = it >> 3, col = it & 7;
      union { uint2 u; u16 h[4]; } dh, dl, ah, al;
      const float dv[4] = {wreg[n].x, wreg[n].y, wreg[n].z, wreg[n].w};
      const float av[4] = {wreg[4 + n].x, wreg[4 + n].y, wreg[4 + n].z, wreg[4 + n].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        split_bf16(dv[e], dh.h[e], dl.h[e]);
        split_bf16(av[e], ah.h[e], al.h[e]);
      }
      const int o = row * GEO_ROW + (((col >> 1) ^ geo_swz(row)) * 8) + (col & 1) * 4;
      *reinterpret_cast<uint2 *>(lds + OFF_WDH + o) = dh.u;
      *reinterpret_cast<uint2 *>(lds + OFF_WDL + o) = dl.u;
      *reinterpret_cast<uint2 *>(lds + OFF_WAH + o) = ah.u;
      *reinterpret_cast<uint2 *>(lds + OFF_WAL + o) = al.u;
    }
  };

  f32x4 accd[8], acca[3][8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    accd[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) acca[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  wload(0);
  for (int ks = 0; ks < GEO_C / 32; ++ks) {
    // ---- phase 1: build shared operands of this k-step --------------------------------------------
    wstore();
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      union { uint4 u; u16 h[8]; } hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float w = div_term[ks * 16 + item_q[n] * 4 + j];
        float sn, cs;
        fast_sincos(xval[n] * w, sn, cs);                          // same fp32 product as the reference (:275)
        split_bf16(sn, hi.h[2 * j], lo.h[2 * j]);                  // interleaved [sin w, cos w] layout (:279-280)
        split_bf16(cs, hi.h[2 * j + 1], lo.h[2 * j + 1]);
      }
      const int o = (item_e[n] * GEO_PAIRS + item_p[n]) * GEO_ROW + (item_q[n] ^ geo_swz(item_p[n])) * 8;
      *reinterpret_cast<uint4 *>(lds + OFF_AH + o) = hi.u;
      *reinterpret_cast<uint4 *>(lds + OFF_AL + o) = lo.u;
    }
    __syncthreads();
    if (ks + 1 < GEO_C / 32) wload(ks + 1);                        // next slice flies under the MFMAs
    // ---- phase 2: 96 MFMAs per wave ------------------------------------------------------------------
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = (e * GEO_PAIRS + mt * 16 + c) * GEO_ROW + (g ^ geo_swz(c)) * 8;
      ah[e] = *reinterpret_cast<const bf16x8 *>(lds + OFF_AH + o);
      al[e] = *reinterpret_cast<const bf16x8 *>(lds + OFF_AL + o);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int o = (nh * 128 + t * 16 + c) * GEO_ROW + (g ^ geo_swz(c)) * 8;
      const bf16x8 wdh = *reinterpret_cast<const bf16x8 *>(lds + OFF_WDH + o);
      const bf16x8 wdl = *reinterpret_cast<const bf16x8 *>(lds + OFF_WDL + o);
      const bf16x8 wah = *reinterpret_cast<const bf16x8 *>(lds + OFF_WAH + o);
      const bf16x8 wal = *reinterpret_cast<const bf16x8 *>(lds + OFF_WAL + o);
      accd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], wdh, accd[t], 0, 0, 0);
      accd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wdh, accd[t], 0, 0, 0);
      accd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], wdl, accd[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        acca[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1 + k], wah, acca[k][t], 0, 0, 0);
        acca[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1 + k], wah, acca[k][t], 0, 0, 0);
        acca[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1 + k], wal, acca[k][t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: + biases, max over the k angular neighbours, one write --------------------------------
  // A = pairs (rows), B = outputs (cols): C layout row = pair g*4+r, col = output c
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int n = nh * 128 + t * 16 + c;
    const float bias = bd[n] + ba[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long pr = pair0 + mt * 16 + g * 4 + r;
      const float v = accd[t][r] + fmaxf(fmaxf(acca[0][t][r], acca[1][t][r]), acca[2][t][r]) + bias;
      if (pr < NP) {
        if (HALF) reinterpret_cast<_Float16 *>(outv)[pr * GEO_C + n] = (_Float16)v;
        else reinterpret_cast<float *>(outv)[pr * GEO_C + n] = v;
      }
    }
  }
}

// ---- round 6: the pre-split form without the operand phase ---------------------------------------------
// geo_embed_kernel above spends a k-step in two phases behind two barriers: every thread builds a share of the sinusoid fragments and
// of the weight slices in LDS (12 ds_write_b128 per thread), then every wave reads 40 fragments back for 96 matrix instructions;
// nothing of phase 1 runs under phase 2 (one workgroup per CU: 120 KB of LDS), and the counted "one bank conflict per matrix
// instruction" is the 2-way conflict of those twelve stores (8 extra LDS cycles each = 96 per wave and k-step).  Here
//   * a lane builds ITS OWN sinusoid fragments in registers: lane (c, g) of wave (mt, nh) needs, per embedding, the 8 channels
//     [8 g, 8 g + 8) of pair 16 mt + c = four sin / cos pairs; the NEXT k-step's are computed while this k-step's matrix
//     instructions issue.  No LDS traffic, no barrier for them (the two nh waves of a pair tile compute the same values twice:
//     16 sincos per lane and k-step instead of 8, on vector slots the matrix pipe leaves idle);
//   * LDS holds only the weight slices, in two stages of [4 parts][256 rows][32] bf16 = 64 KB: slice ks + 1 is stored while slice ks
//     is multiplied, ONE barrier per k-step.  Rows are 64 bytes apart, the chunk of a row XOR-ed with 2 (row >> 3 & 1): the sixteen
//     lanes of a ds_read_b128 group ({c 0-3, 12-15 at chunk g} + {c 4-11 at chunk g + 1}) then cover the sixteen 16-byte slots of
//     the 256-byte bank row exactly once, and a ds_write_b128 group (two rows x four chunks) is 128 contiguous bytes.
// Arithmetic: the same fragments, the same products in the same order per accumulator -> the same bits as geo_embed_kernel
// (tests/test_gpu_pose.py::test_half_stored_geo_embedding_and_its_reader holds the two to torch.equal).
// MEASURED (profiles/r06_geo_embed.md): LDS bank conflicts 78.2 M -> 0, LDS-array cycles 287 M -> 104 M, waits on LDS 123 M -> 9 M per
// launch -- and 1.95 ms against 1.90 ms for the two-phase kernel at 32 instances (the PEM stage 20.97 against 20.83 ms).  Ablation
// builds (S6D_G2_ABL): without the weight staging 1.66 ms, without the sinusoids 1.67, without both 1.41 -- and the BARE stream
// of matrix instructions, nothing else in the loop, 1.41 ms = 0.56 of the nominal bf16 rate: the chip sustains ~1400 TFLOP/s of
// 16x16x32 products under its power limit, everything beside them costs clock, and this form spends more vector instructions
// (each nh wave builds the fragments its twin builds too: 3.7 against 1.7 per matrix instruction) than it saves in LDS work.
// The two-phase kernel therefore stays the default (s6d_set_geo_embed_form(2) selects this one).
#ifndef S6D_G2_ABL
#define S6D_G2_ABL 0                                               // timing-only ablation builds (tools/probes/geo_variants.sh): 1 no weight staging, 2 no sinusoids, 4 one fragment set
#endif
constexpr int G2_WROW = 32;                                        // bf16 per row and part in a stage (64 B)
constexpr int G2_PART = GEO_C * G2_WROW;
constexpr int G2_STAGE = 4 * G2_PART;                              // W_d hi | W_d lo | W_a hi | W_a lo
constexpr int G2_LDS_BYTES = 2 * G2_STAGE * 2;                     // 128 KB
#define G2_LDS(T) __attribute__((address_space(3))) T
#define G2_GLOBAL(T) __attribute__((address_space(1))) T
#ifdef HIPEMU
#define G2_VMCNT0() hipemu::vmcnt_wait(0)
#else
#define G2_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
__device__ __forceinline__ int g2_swz(int row) { return ((row >> 3) & 1) << 1; }

template <bool HALF>
__global__ __launch_bounds__(GEO_THREADS) void geo_embed2_kernel(const float *__restrict__ idx4, long NP, const u16 *__restrict__ Wd,
                                                                const float *__restrict__ bd, const u16 *__restrict__ Wa,
                                                                const float *__restrict__ ba, const float *__restrict__ div_term,
                                                                void *__restrict__ outv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u16 *lds = reinterpret_cast<u16 *>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int mt = wave & 3, nh = wave >> 2;
  const long pair0 = (long)blockIdx.x * GEO_PAIRS;
  float xv[4];
  {
    const long pr = min(pair0 + mt * 16 + c, NP - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = idx4[pr * 4 + e];
  }
  // weight slice of k-step ks -> stage: LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write).  A piece = one part's 16 rows
  // x 64 B = 1 KiB per wave instruction; lane l of wave w lands at element it = 64 w + l (+ 512 for the second piece) of the part's
  // [256 rows][4 chunks] image, i.e. row it / 4 at chunk POSITION it % 4, and fetches the chunk (position ^ swizzle(row)) there.
  auto wstage = [&](int ks, int stage) __attribute__((always_inline)) {
    const u16 *parts[4] = {Wd, Wd + GEO_C * GEO_C, Wa, Wa + GEO_C * GEO_C};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int it = tid + n * GEO_THREADS;
        const int row = it >> 2, ch = it & 3;
        const u16 *src = parts[q] + (size_t)row * GEO_C + ks * 32 + ((ch ^ g2_swz(row)) * 8);
        G2_LDS(char) *dst = (G2_LDS(char) *)smem + ((stage * G2_STAGE + q * G2_PART) * 2 + (wave * 64 + n * GEO_THREADS) * 16);
        __builtin_amdgcn_global_load_lds((const G2_GLOBAL(void) *)src, dst, 16, 0, 0);
      }
  };
  // the lane's sinusoid fragments of k-step ks: [embedding] chunk g = frequencies 16 ks + 4 g + j, interleaved [sin, cos] (:279-280)
  auto frags = [&](int ks, bf16x8 (&ah)[4], bf16x8 (&al)[4]) __attribute__((always_inline)) {
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = div_term[ks * 16 + g * 4 + j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      union { bf16x8 v; u16 h[8]; } hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float sn, cs;
        fast_sincos(xv[e] * w[j], sn, cs);                         // same fp32 product as the reference (:275)
        split_bf16(sn, hi.h[2 * j], lo.h[2 * j]);
        split_bf16(cs, hi.h[2 * j + 1], lo.h[2 * j + 1]);
      }
      ah[e] = hi.v;
      al[e] = lo.v;
    }
  };

  f32x4 accd[8], acca[3][8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    accd[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) acca[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ah[4], al[4];
  wstage(0, 0);
  frags(0, ah, al);
  float wq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wq[j] = div_term[16 + g * 4 + j];
  G2_VMCNT0();
  __syncthreads();
#pragma unroll 1
  for (int ks = 0; ks < GEO_C / 32; ++ks) {
    const int st = ks & 1;
    // (no conditions on the last k-step: its surplus slice goes to a stage nobody reads again, its surplus fragments are dropped)
#if !(S6D_G2_ABL & 1)
    wstage(min(ks + 1, GEO_C / 32 - 1), st ^ 1);                   // every wave is past its reads of that stage (barrier below)
#endif
    // the NEXT k-step's sinusoid fragments are built in eight portions (two sin / cos pairs each), one behind each output tile's
    // twelve matrix instructions, with a scheduling fence between the portions: left to itself the compiler runs all ~240 vector
    // instructions first and the 96 products after them, and both waves of a SIMD then want the same pipe at the same time;
    // in portions the two waves fall one portion out of step and one's vector work runs under the other's products
    float wq2[4];                                                  // the frequencies of the portions of the NEXT iteration (a load
#pragma unroll                                                     // consumed in this one would stall the first portion)
    for (int j = 0; j < 4; ++j) wq2[j] = div_term[min(ks + 2, GEO_C / 32 - 1) * 16 + g * 4 + j];
    union { bf16x8 v; u16 h[8]; } nhi[4], nlo[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int o = st * G2_STAGE + (nh * 128 + ((S6D_G2_ABL & 4) ? 0 : t) * 16 + c) * G2_WROW + (g ^ g2_swz(c)) * 8;
      const bf16x8 wdh = *reinterpret_cast<const bf16x8 *>(lds + o);
      const bf16x8 wdl = *reinterpret_cast<const bf16x8 *>(lds + G2_PART + o);
      const bf16x8 wah = *reinterpret_cast<const bf16x8 *>(lds + 2 * G2_PART + o);
      const bf16x8 wal = *reinterpret_cast<const bf16x8 *>(lds + 3 * G2_PART + o);
      // issue order: the four accumulators of the tile in turn, so that a matrix instruction's accumulator input is three
      // instructions old (a dependent 16x16x32 issued back to back waits for the whole pipe: the bare product stream of this
      // kernel measured 0.56 of the peak rate with the three terms of an accumulator in a row); per accumulator the order of the
      // terms -- x_hi w_hi, x_lo w_hi, x_hi w_lo -- is unchanged, so the bits are
      accd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], wdh, accd[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) acca[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1 + k], wah, acca[k][t], 0, 0, 0);
      accd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[0], wdh, accd[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) acca[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[1 + k], wah, acca[k][t], 0, 0, 0);
      accd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[0], wdl, accd[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) acca[k][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[1 + k], wal, acca[k][t], 0, 0, 0);
#if !(S6D_G2_ABL & 2)
      {
        const int e = t >> 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = (t & 1) * 2 + jj;
          float sn, cs;
          fast_sincos(xv[e] * wq[j], sn, cs);                      // same fp32 product as the reference (:275)
          split_bf16(sn, nhi[e].h[2 * j], nlo[e].h[2 * j]);        // interleaved [sin w, cos w] layout (:279-280)
          split_bf16(cs, nhi[e].h[2 * j + 1], nlo[e].h[2 * j + 1]);
        }
      }
#else
      nhi[t >> 1].v = ah[t >> 1];
      nlo[t >> 1].v = al[t >> 1];
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 nh_[4], nl_[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      nh_[e] = nhi[e].v;
      nl_[e] = nlo[e].v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) wq[j] = wq2[j];
    G2_VMCNT0();                                                   // the next slice has landed (issued ~1500 matrix-pipe cycles ago)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ah[e] = nh_[e];
      al[e] = nl_[e];
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int n = nh * 128 + t * 16 + c;
    const float bias = bd[n] + ba[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long pr = pair0 + mt * 16 + g * 4 + r;
      const float v = accd[t][r] + fmaxf(fmaxf(acca[0][t][r], acca[1][t][r]), acca[2][t][r]) + bias;
      if (pr < NP) {
        if (HALF) reinterpret_cast<_Float16 *>(outv)[pr * GEO_C + n] = (_Float16)v;
        else reinterpret_cast<float *>(outv)[pr * GEO_C + n] = v;
      }
    }
  }
}

}  // namespace s6d

using namespace s6d;

static int g_geo_form = 1;          // 1: geo_embed_kernel<., true> (default: 2 % ahead in the step); 2: geo_embed2_kernel (selectable; profiles/r06_geo_embed.md)
extern "C" int s6d_set_geo_embed_form(int form) {
  if (form != 1 && form != 2) return S6D_EINVAL;
  g_geo_form = form;
  return S6D_OK;
}

template <bool HALF, bool PRE>
static void geo_launch_t(const float *idx4, long NP, const float *Wd, const float *bd, const float *Wa, const float *ba,
                         const float *div_term, void *out, void *stream) {
  const size_t lds = (size_t)GEO_LDS_ELEMS * 2;
  const unsigned grid = (unsigned)((NP + GEO_PAIRS - 1) / GEO_PAIRS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&geo_embed_kernel<HALF, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((geo_embed_kernel<HALF, PRE>), dim3(grid), dim3(GEO_THREADS), lds, as_stream(stream), idx4, NP, Wd, bd, Wa, ba, div_term, out);
}

static int geo_launch(const float *idx4, long NP, const void *Wd, const float *bd, const void *Wa, const float *ba,
                      const float *div_term, int C, int K, void *out, bool half, bool pre, void *stream) {
  if (NP < 0) return S6D_EINVAL;
  if (C != GEO_C || K != 3) return S6D_EUNSUPPORTED;      // released model: hidden_dim 256, angle_k 3
  if (NP == 0) return S6D_OK;
  if (!idx4 || !Wd || !bd || !Wa || !ba || !div_term || !out) return S6D_EINVAL;
  if (pre && (((uintptr_t)Wd | (uintptr_t)Wa) & 15)) return S6D_EINVAL;
  const float *wd = reinterpret_cast<const float *>(Wd), *wa = reinterpret_cast<const float *>(Wa);
  if (pre && g_geo_form == 2) {
    const unsigned grid = (unsigned)((NP + GEO_PAIRS - 1) / GEO_PAIRS);
    const u16 *wdh = reinterpret_cast<const u16 *>(Wd), *wah = reinterpret_cast<const u16 *>(Wa);
    if (half) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&geo_embed2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
      hipLaunchKernelGGL((geo_embed2_kernel<true>), dim3(grid), dim3(GEO_THREADS), G2_LDS_BYTES, as_stream(stream), idx4, NP, wdh, bd, wah, ba, div_term, out);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&geo_embed2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
      hipLaunchKernelGGL((geo_embed2_kernel<false>), dim3(grid), dim3(GEO_THREADS), G2_LDS_BYTES, as_stream(stream), idx4, NP, wdh, bd, wah, ba, div_term, out);
    }
    return launch_status();
  }
  if (half) {
    if (pre) geo_launch_t<true, true>(idx4, NP, wd, bd, wa, ba, div_term, out, stream);
    else geo_launch_t<true, false>(idx4, NP, wd, bd, wa, ba, div_term, out, stream);
  } else {
    if (pre) geo_launch_t<false, true>(idx4, NP, wd, bd, wa, ba, div_term, out, stream);
    else geo_launch_t<false, false>(idx4, NP, wd, bd, wa, ba, div_term, out, stream);
  }
  return launch_status();
}

extern "C" int s6d_geo_embedding_f32(const float *idx4, long NP, const float *Wd, const float *bd, const float *Wa,
                                     const float *ba, const float *div_term, int C, int K, float *out, void *stream) {
  return geo_launch(idx4, NP, Wd, bd, Wa, ba, div_term, C, K, out, false, false, stream);
}

extern "C" int s6d_geo_embedding_f16(const float *idx4, long NP, const float *Wd, const float *bd, const float *Wa,
                                     const float *ba, const float *div_term, int C, int K, void *out_f16, void *stream) {
  return geo_launch(idx4, NP, Wd, bd, Wa, ba, div_term, C, K, out_f16, true, false, stream);
}

extern "C" int s6d_geo_embedding_split(const float *idx4, long NP, const void *Wd_hilo, const float *bd, const void *Wa_hilo,
                                       const float *ba, const float *div_term, int C, int K, void *out, int out_f16, void *stream) {
  return geo_launch(idx4, NP, Wd_hilo, bd, Wa_hilo, ba, div_term, C, K, out, out_f16 != 0, true, stream);
}
