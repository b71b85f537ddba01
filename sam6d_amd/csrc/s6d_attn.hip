// Fused ViT attention with decomposed relative-position bias for the SAM image encoder (gfx950).
//
// Reference: segment_anything/modeling/image_encoder.py
//   Block.forward :166-182 (norm1 -> window_partition(pad 64->70) -> attn -> unpartition),
//   Attention.forward :224-240, add_decomposed_rel_pos :325-361, get_rel_pos :292-322.
// The reference materialises (B*25*16,196,196) / (B*16,4096,4096) fp32 score tensors plus two
// partition copies per block.  Here one kernel reads the qkv GEMM output of the 4096 REAL tokens,
// forms windows by address arithmetic (out-of-image window slots take the qkv bias as q/k/v:
// quirk Q2, padded tokens are zeros after norm1 and DO act as keys), runs flash-style online
// softmax on MFMA tiles and writes the attended tokens straight into the (B,H,W,C) map.
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16, wave64):
//   S^T tile (16 keys x 16 queries) = K_tile (A: lane -> key l&15, 8 contiguous d) x Q^T (B: lane ->
//   query l&15, 8 contiguous d): the C layout then gives each lane 4 keys x 1 QUERY (col = l&15), so
//   softmax statistics are per-lane scalars (+2 shuffles across the 4 lane groups).
//   O^T tile (16 d x 16 queries)   = V^T (A, from a transposed LDS image) x P^T (B): P^T fragments
//   are exactly the exponentiated S^T registers (k-index permuted consistently on both operands),
//   so P never leaves registers and O^T keeps the per-lane query -> rescaling needs no shuffles.
//   The decomposed bias  rel_h[q,ky] + rel_w[q,kx]  comes from two per-query tables
//   T[j][q] = rel_pos[j] . q  (2S-1 entries) built with the same MFMA shape in the strip prologue.
#include "s6d_common.h"
#include <stdlib.h>

// Element type of q / k / v / P / the output.  The file is compiled twice: as is for bf16 (SAM, DINOv2: `bf16x8`, `f2bf` mean what
// they say), and from csrc/s6d_attn_f16.hip with S6D_ATTN_F16 = 1 for IEEE half (the PEM's ViT-B, round 3): the same kernels in
// namespace s6d_h with v_mfma_f32_16x16x32_f16, where the names below stand for the half forms and only s6d_seq_attention_f16
// is exported.  Nothing else in the file depends on the element's bit layout (probabilities are <= 1, sums are fp32).
#ifndef S6D_ATTN_F16
#define S6D_ATTN_F16 0
#endif
#if S6D_ATTN_F16
#define S6D_ATTN_NS s6d_h
#define S6D_ATTN_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define S6D_ATTN_ONE 0x3C00
#else
#define S6D_ATTN_NS s6d
#define S6D_ATTN_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define S6D_ATTN_ONE 0x3F80
#endif

namespace S6D_ATTN_NS {
using namespace s6d;

#if S6D_ATTN_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

#if S6D_ATTN_F16
__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even
  union { _Float16 b; u16 u; } x;
  x.b = (_Float16)f;
  return x.u;
}
__device__ __forceinline__ float bf2f(u16 h) {
  union { _Float16 b; u16 u; } x;
  x.u = h;
  return (float)x.b;
}
#else
__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
  union { __bf16 b; u16 u; } x;
  x.b = (__bf16)f;
  return x.u;
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
#endif

struct AttnParams {
  const u16 *qkv;      // (B,H,W,3,nh,HD) bf16
  const u16 *qkv_bias; // (3,nh,HD) bf16  (q/k/v of out-of-image window slots)
  const u16 *rel_h;    // (2S-1,HD) bf16 or nullptr
  const u16 *rel_w;    // (2S-1,HD) bf16
  u16 *out;            // (B,H,W,nh,HD) bf16
  int B, H, W, nh;
  int ws;              // window size (0 = global attention over the H x W grid)
  int S;               // side of the attention grid (ws, or H for global)
  int T;               // tokens per attention problem = S*S
  int nwx, nwy;        // windows per image along x / y (1,1 for global)
  int LT;              // padded table length (multiple of 16, >= 2S-1)
  unsigned magicS;     // ceil(2^32 / S): n / S == umulhi(n, magicS) for n < 2^32 / S
  float scale_log2;    // softmax scale * log2(e): scores live in the exp2 domain
  // q/k/v element (token, which, head, d) sits at qkv + token * tok_stride + which * which_stride + head * head_stride + d.
  //   token-major (the raw Linear output (B,H,W,3,nh,HD)): tok_stride = 3 nh HD, which_stride = nh HD, head_stride = HD
  //   head-major  ((3, nh, B H W, HD), written by the qkv GEMM's column-block epilogue): tok_stride = HD, head_stride = B H W HD,
  //               which_stride = nh B H W HD -- a head's rows of one window row / key tile are contiguous whole lines
  long tok_stride, which_stride, head_stride;
};

__device__ __forceinline__ const u16 *qkv_at(const AttnParams &p, size_t tok, int which, int head) {
  return p.qkv + tok * (size_t)p.tok_stride + (size_t)which * (size_t)p.which_stride + (size_t)head * (size_t)p.head_stride;
}

// Ablation switches for profiling are COMPILE-time (-DS6D_ATTN_ABLATE=mask through S6D_EXTRA_HIPCC_FLAGS): 1 = no K/V
// loads, 2 = no tile math, 4 = no softmax arithmetic, 8 = no PV, 16 = no QK^T, 32 = no LDS stores of the staged tile.  As run-time branches they cut the
// tile loop into a dozen basic blocks and kept the scheduler from placing MFMAs beside the softmax VALU work.
#ifndef S6D_ATTN_ABLATE
#define S6D_ATTN_ABLATE 0
#endif
constexpr int kAbl = S6D_ATTN_ABLATE;
// Global-attention layout / schedule switches (defaults = what is measured fastest; tools/attn_variants.sh builds the others):
//   S6D_GLB_KSWZ  K image rows whose (row >> 2 ^ row >> 3) & 1 is set keep their 16-byte chunks pairwise swapped: with an odd row
//                 stride (13 chunks) the two row sets of a ds_read_b128 lane group ({0-3,12-15} reading chunk g, {4-11} reading
//                 chunk g + 1) otherwise meet on 5 of 16 bank slots (2-way conflict on every K fragment read)
//   S6D_GLB_THLD  row stride (floats) of the per-query th tables: at 64 the 16 queries of a lane group read ONE bank
//   S6D_GLB_PRIO  s_setprio 1 around the MFMA phases of a tile (the wave in its MFMA phase wins issue over the one in softmax)
//   S6D_GLB_WAVES waves per workgroup (4: two workgroups per CU; 8: one, each staged K/V tile shared by twice the queries)
#ifndef S6D_GLB_KSWZ
#define S6D_GLB_KSWZ 1
#endif
#ifndef S6D_GLB_THLD
#define S6D_GLB_THLD 65
#endif
#ifndef S6D_GLB_PRIO
#define S6D_GLB_PRIO 1
#endif
#ifndef S6D_WIN16_ASM_DMA
#define S6D_WIN16_ASM_DMA 1          // persistent window kernel: LDS-DMA as inline asm (1) or through the builtin (0).  With the builtin hipcc
#endif                               // waits vmcnt(0) in front of the next LDS read (bias tables; first V fragment): the prefetch of the NEXT
                                     // item's images did not overlap the current item's arithmetic at all (ISA read in round 4)
#ifndef S6D_WIN16_QDEFER
#define S6D_WIN16_QDEFER 1           // the next item's Q fragments are loaded half way through the PV pass and MASKED at the start of the
#endif                               // next item (1) instead of right behind the loads (0: a full fetch latency inside the PV pass)
#ifndef S6D_WIN16_STREAM
#define S6D_WIN16_STREAM 1           // 14 x 14 windows: one streaming pass without the exact maximum (win16_pass_stream), exact pass as the fallback
#endif
#ifndef S6D_WIN16_KSWZ
#define S6D_WIN16_KSWZ 1             // the same chunk swizzle on the persistent window kernel's compact K image (11-chunk rows).  Round 2 measured
                                     // it at 0.265 against 0.260 ms (the swizzled source computed per DMA instruction); since round 5 the per-lane
                                     // chunk map is made once per kernel and the swizzle is free: 0.2403 -> 0.2372 ms (profiles/r05_attn_window_stream.txt)
#endif
#ifndef S6D_GLB_WAVES
#define S6D_GLB_WAVES 4
#endif
__device__ __forceinline__ int kswz(int row) { return ((row >> 2) ^ (row >> 3)) & 1; }
// S6D_G64_TIMING (probe build; writes 320 bytes BEHIND the output tensor, which tools/attn_time.py allocates): shader-clock totals per phase of attn_global64_kernel's tile loop,
// per wave of workgroup 0: [barrier wait | DMA issue | QK^T + scale + max | exp + pack | P V]  (tools/attn_time.py prints them)
#ifndef S6D_G64_TIMING
#define S6D_G64_TIMING 0
#endif
#if S6D_G64_TIMING
#define S6D_TICK(tk, i)                                  \
  do {                                                   \
    __builtin_amdgcn_sched_barrier(0);                   \
    (tk)[i] = (long long)__builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#else
#define S6D_TICK(tk, i) do { } while (0)
#endif

constexpr float kLog2e = 1.4426950408889634f;
// v_exp_f32 without the denormal-range fix-up of fast_exp2(): arguments here are <= 2^kDefer and tiny results may flush
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define S6D_LDS(T) __attribute__((address_space(3))) T

template <int HD>
struct Cfg {
  static constexpr int KS = (HD + 31) / 32;     // k-steps of 32 over the head dim (zero padded)
  static constexpr int HDP = KS * 32;           // padded head dim
  static constexpr int DT = HD / 16;            // 16-wide d tiles of the output
  static constexpr int KROW = HDP + 8;          // K image row stride (bf16 elements): +16 B pad
  // V image row stride (row-major [key][d]): an ODD multiple of 16 elements (32 B), so the 8 key rows that one
  // half-wave of a ds_read_b64_tr_b16 touches start on 8 distinct 32-byte bank groups (HD + 8 was 2-way conflicted)
  // Head dim 64 takes HD + 8 instead (144-byte rows: the eighth row of a half-wave wraps onto the first one's first four banks, a
  // 2-way conflict on one row in eight) -- at 160 bytes the K / V images of a 257-token sequence miss the 80 KiB that lets TWO
  // workgroups share a CU by 3 KiB (round 4, attn_window_kernel's compact layout).
  static constexpr int VROW = HD == 64 ? HD + 8 : (((HD + 15) / 16) | 1) * 16;
  static constexpr int KT = 64;                 // keys per tile
  static constexpr int KPARTS = HDP / 8, VPARTS = HD / 8;
};

__device__ __forceinline__ int div_S(const AttnParams &p, int n) { return (int)__umulhi((unsigned)n, p.magicS); }

// global token index of slot `t` of problem (b, wy, wx); false for an out-of-image / out-of-range slot
// (`off` is always a valid token index -- clamped -- so loads through it may be issued unconditionally)
__device__ __forceinline__ bool token_offset(const AttnParams &p, int b, int wy, int wx, int t, size_t &off) {
  if (p.ws == 0) {                      // global attention: slots are the image tokens in raster order
    off = (size_t)b * p.T + min(t, p.T - 1);
    return t < p.T;
  }
  const int ty = div_S(p, t), tx = t - ty * p.S;
  const int y = wy * p.ws + ty, x = wx * p.ws + tx;
  off = ((size_t)(b * p.H + min(y, p.H - 1)) * p.W + min(x, p.W - 1));
  return (y < p.H) && (x < p.W) && (t < p.T);
}

// 8 consecutive head-dim elements [d0, d0+8) of q/k/v (which = 0/1/2) for a token slot.  Branch-free on
// purpose: the load is unconditional (clamped address, pointer select), zeros are selected afterwards, so the
// compiler can keep a whole batch of these in flight instead of one exec-masked load + vmcnt(0) each.
// `tok` must be a valid token index even when !valid (callers clamp).
template <int HD>
__device__ __forceinline__ uint4 load_chunk(const AttnParams &p, int which, int head, bool valid, size_t tok, int d0) {
  const int C = p.nh * HD;
  const int dc = d0 < HD ? d0 : HD - 8;
  const u16 *a = qkv_at(p, tok, which, head) + dc;
  const u16 *bsrc = p.qkv_bias + (size_t)which * C + head * HD + dc;
  const uint4 v = *reinterpret_cast<const uint4 *>(valid ? a : bsrc);
  return d0 < HD ? v : make_uint4(0, 0, 0, 0);
}

// MODE 0: bias from per-query LDS tables th/tw (any S);  MODE 1: aligned fast path (S == 64, one key row per
// tile): th value is one LDS word per tile, tw lives in 16 registers;  MODE 2: no positional bias.
//   Kl : K image of the tile [64][KROW];  Vl : V image of the tile [64][VROW] (row-major, read with
//   ds_read_b64_tr_b16: lane c of a 16-lane group receives column c of a 4-key block).
// NS query strips (16 queries each) are processed against the same K / V fragments: every LDS operand read
// feeds NS MFMAs.
template <int HD, int NS>
struct StripState {
  bf16x8 qf[NS][Cfg<HD>::KS];
  float twr[NS][16];
  float m_run[NS];
  f32x4 lacc[NS];                          // running row sums: every element of lacc[n] is the sum for query lane & 15
  f32x4 oacc[NS][Cfg<HD>::DT];
  const float *th[NS], *tw[NS];
  int qy[NS], qx[NS];
};

// SUBS: 16-key sub-tiles of this tile that exist (a sequence's tail tile: 257 keys = 4 tiles + 1 sub-tile): the score MFMAs of the
// others are skipped (their probabilities are 0) and the PV pass covers ceil(SUBS / 2) 32-key steps; the V image must hold finite
// values for the absent keys of the last step.
template <int HD, int MODE, int NS, bool KSWZ = false, bool PRIO = false, int SUBS = 4>
__device__ __forceinline__ void process_tile(const AttnParams &p, const u16 *Kl, const u16 *Vl, int key0,
                                             StripState<HD, NS> &st, const float (&thv)[NS], int lane, long long *tk = nullptr) {
  using C = Cfg<HD>;
  static_assert(SUBS >= 1 && SUBS <= 4, "a tile has four 16-key sub-tiles");
  const int g = lane >> 4, c = lane & 15;
  const int gk = KSWZ ? (g ^ kswz(c)) : g;               // chunk of this lane's K fragment inside its group of 4 (see S6D_GLB_KSWZ)
  float s[NS][4][4];
  if (PRIO) __builtin_amdgcn_s_setprio(1);
  auto kfrag = [&](int sub, int ks) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8 *>(Kl + (sub * 16 + c) * C::KROW + ks * 32 + gk * 8);
  };
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    f32x4 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (sub >= SUBS) {
#pragma unroll
      for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[n][sub][r] = -1e30f;
      continue;
    }
    if (!(kAbl & 16)) {
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const bf16x8 a = kfrag(sub, ks);
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[n] = S6D_ATTN_MFMA16(a, st.qf[n][ks], acc[n]);
      }
    }
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[n][r] * p.scale_log2;
        if (MODE == 1) {
          s[n][sub][r] = v + st.twr[n][sub * 4 + r];      // thv (constant over the tile for this lane) joins the max only
        } else {
          const int kk = key0 + sub * 16 + g * 4 + r;     // key slot of this score; query = lane & 15
          if (MODE == 0) {
            const int ky = div_S(p, kk), kx = kk - ky * p.S;
            const int jh = min(max(st.qy[n] - ky + p.S - 1, 0), p.LT - 1), jw = st.qx[n] - kx + p.S - 1;
            v += st.th[n][c * p.LT + jh] + st.tw[n][c * p.LT + jw];
          }
          s[n][sub][r] = kk < p.T ? v : -1e30f;
        }
      }
  }
  union PB { bf16x8 v; u16 h[8]; };
  PB pb[NS][2];
  union VA { bf16x8 v; s16x4 q[2]; };
  if (PRIO) __builtin_amdgcn_s_setprio(0);
  if (kAbl & 4) {                                          // ablation: no softmax arithmetic
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[n][sub >> 1].h[(sub & 1) * 4 + r] = f2bf(s[n][sub][r]);
  } else {
    // deferred rescaling: keep the old running max while no query's max grew by more than 2^kDefer (P stays
    // <= 2^kDefer, exact in bf16's exponent range); the O / l rescale is skipped on those tiles.  Both strips'
    // decisions are taken first, so the (rare) rescale is ONE wave-uniform branch and the rest of the tile is
    // straight-line code the scheduler can interleave with the MFMAs.
    constexpr float kDefer = 6.0f;
    float mx[NS];
    bool grow[NS], any_grow = false;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      float v = s[n][0][0];
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaxf(v, s[n][sub][r]);
      v = fmaxf(v, __shfl_xor(v, 16));
      v = fmaxf(v, __shfl_xor(v, 32));
      if (MODE == 1) v += thv[n];                             // true score maximum of the tile
      mx[n] = v;
      grow[n] = __any(v - st.m_run[n] > kDefer);
      any_grow |= grow[n];
    }
    S6D_TICK(tk, 3);
    if (any_grow) {                                           // wave-uniform
#pragma unroll
      for (int n = 0; n < NS; ++n)
        if (grow[n]) {
          const float m_new = fmaxf(st.m_run[n], mx[n]);
          const float alpha = fast_exp2(st.m_run[n] - m_new);
          st.lacc[n] *= alpha;
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) st.oacc[n][dt] *= alpha;
          st.m_run[n] = m_new;
        }
    }
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const float m_sub = (MODE == 1) ? st.m_run[n] - thv[n] : st.m_run[n];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pb[n][sub >> 1].h[(sub & 1) * 4 + r] = sub < SUBS ? f2bf(fast_exp2(s[n][sub][r] - m_sub)) : (u16)0;
    }
  }
  // O^T += V^T P^T: k-step j covers keys [32j, 32j+32); MFMA k-index e<4 -> key 32j+g*4+e, e>=4 -> 32j+16+g*4+(e-4).
  // The row sums ride the matrix core too: an all-ones A fragment makes every row of lacc the sum over keys of the
  // SAME bf16-rounded P that multiplies V (no VALU adds, no lane exchange).
  if (kAbl & 8) {                                          // ablation: no PV (keep P live)
#pragma unroll
    for (int n = 0; n < NS; ++n) st.lacc[n][0] += (float)pb[n][0].h[0] + (float)pb[n][1].h[7];
    return;
  }
  union { bf16x8 v; u16 h[8]; } ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones.h[i] = S6D_ATTN_ONE;            // 1.0 in the element type
  S6D_TICK(tk, 4);
  if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int j = 0; j < (SUBS + 1) / 2; ++j) {
    const u16 *vrow = Vl + (32 * j + g * 4 + (c >> 2)) * C::VROW + (c & 3) * 4;
#pragma unroll
    for (int n = 0; n < NS; ++n) st.lacc[n] = S6D_ATTN_MFMA16(ones.v, pb[n][j].v, st.lacc[n]);
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      VA va;
      va.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + dt * 16));
      va.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + 16 * C::VROW + dt * 16));
#pragma unroll
      for (int n = 0; n < NS; ++n)
        st.oacc[n][dt] = S6D_ATTN_MFMA16(va.v, pb[n][j].v, st.oacc[n][dt]);
    }
  }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}

// Range guard of the passes that keep a FIXED reference instead of a running maximum (process_tile_nomax, win16_pass_stream): a row
// is redone (second pass / raised reference) when its sum of P = exp2(s - m) is not below 2^100 (inf / NaN included) OR when one of
// its O^T accumulators is not finite (the accumulators hold sum P V, i.e. up to 2^100 |V|, and overflowed float32 for
// |V| above 2^28 while the sum still passed; tests/test_gpu_attn.py runs |V| = 2^50 under P up to 2^90).  A lower sum limit instead
// (2^60 was tried) sends ordinary frames through the second pass: the seeded ViT-H of the benched step has rows 60 .. 100 log2
// units above their first tile, and the global kernel went 1.81 -> 3.25 ms.
constexpr float kNoMaxSumLimit = 1.2676506e30f;                    // 2^100
constexpr float kNoMaxFinite = 3.0e38f;


// ---- round 5: the 64 x 64 global tile WITHOUT a running maximum ---------------------------------------------------------------
// process_tile spends, per score, scale + bias (fma), running max (max / max3), subtract, exp2 and half a pack, plus two
// cross-lane exchanges and a wave vote per strip for the maximum -- 130 VALU + 32 exp2 beside 48 matrix instructions per tile, and
// the maximum is a barrier between ALL score instructions of a tile and ALL its exponentials.  None of the maximum is needed
// after the first tile: P only has to stay representable.  bf16 has float32's exponent range and the accumulators (row sums on
// the matrix core, O^T) are float32, so with m fixed at the FIRST tile's row maximum P = exp2(s - m) is computed to the same
// relative precision as with a running maximum as long as no row's sum approaches 2^127.  So after tile 0 (process_tile, which
// sets m_run):
//   * the rel-pos column bias rides the matrix core: the score chain starts from C = tw / scale_log2 instead of 0;
//   * P = exp2(fma(acc, scale_log2, th - m)): ONE fma per score; no max, no subtract, no exchange, no vote, no rescale branch, and
//     the exponentials of sub-tile s are independent of the score instructions of sub-tile s + 1;
//   * at the end of the tile loop a row sum that is not < kNoMaxSumLimit = 2^100 (inf / NaN included), or a non-finite accumulator, makes the WORKGROUP run its tile loop again
//     with process_tile for every tile (attn_global64_kernel): the old arithmetic is the fallback, so any input the old kernel
//     handled is still handled -- scores that grow by more than 2^100 over the first 64 keys' maximum take the slow path.
template <int HD, int NS, bool KSWZ, bool PRIO>
__device__ __forceinline__ void process_tile_nomax(const AttnParams &p, const u16 *Kl, const u16 *Vl, StripState<HD, NS> &st,
                                                   const f32x4 (&cbias)[NS][4], const float (&nb)[NS], int lane) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
  const int gk = KSWZ ? (g ^ kswz(c)) : g;
  union PB { bf16x8 v; u16 h[8]; };
  PB pb[NS][2];
  if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    f32x4 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) acc[n] = cbias[n][sub];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Kl + (sub * 16 + c) * C::KROW + ks * 32 + gk * 8);
#pragma unroll
      for (int n = 0; n < NS; ++n) acc[n] = S6D_ATTN_MFMA16(a, st.qf[n][ks], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)     // (scalar FMAs on purpose: as two v_pk_fma_f32 per lane the kernel measured 1.6 % slower, profiles/r05_attn_pkfma_ab.txt)
        pb[n][sub >> 1].h[(sub & 1) * 4 + r] = f2bf(fast_exp2(__builtin_fmaf(acc[n][r], p.scale_log2, nb[n])));
  }
  union { bf16x8 v; u16 h[8]; } ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones.h[i] = S6D_ATTN_ONE;
  union VA { bf16x8 v; s16x4 q[2]; };
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const u16 *vrow = Vl + (32 * j + g * 4 + (c >> 2)) * C::VROW + (c & 3) * 4;
#pragma unroll
    for (int n = 0; n < NS; ++n) st.lacc[n] = S6D_ATTN_MFMA16(ones.v, pb[n][j].v, st.lacc[n]);
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      VA va;
      va.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + dt * 16));
      va.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + 16 * C::VROW + dt * 16));
#pragma unroll
      for (int n = 0; n < NS; ++n) st.oacc[n][dt] = S6D_ATTN_MFMA16(va.v, pb[n][j].v, st.oacc[n][dt]);
    }
  }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}

// ---- staging: global -> registers -> LDS, split so the loads can fly under the previous tile's math ------
template <int HD, int THREADS>
struct Stager {
  using C = Cfg<HD>;
  static constexpr int NK = (64 * C::KPARTS + THREADS - 1) / THREADS;
  static constexpr int NV = (64 * C::VPARTS + THREADS - 1) / THREADS;
  uint4 k[NK], v[NV];

  __device__ __forceinline__ void load(const AttnParams &p, int b, int wy, int wx, int head, int key0, int tid) {
#pragma unroll
    for (int n = 0; n < NK; ++n) {
      const int i = min(tid + n * THREADS, 64 * C::KPARTS - 1);     // surplus lanes repeat the last chunk
      const int key = i / C::KPARTS, part = i - key * C::KPARTS;
      size_t tok;
      const bool valid = token_offset(p, b, wy, wx, key0 + key, tok);
      k[n] = load_chunk<HD>(p, 1, head, valid, tok, part * 8);
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = min(tid + n * THREADS, 64 * C::VPARTS - 1);
      const int key = i / C::VPARTS, part = i - key * C::VPARTS;
      size_t tok;
      const bool valid = token_offset(p, b, wy, wx, key0 + key, tok);
      v[n] = load_chunk<HD>(p, 2, head, valid, tok, part * 8);
    }
  }
  // klim / vlim: rows of this tile that the image holds (a compact image ends inside the last tile)
  __device__ __forceinline__ void store(u16 *Kl, u16 *Vl, int tid, int klim = 64, int vlim = 64) const {
#pragma unroll
    for (int n = 0; n < NK; ++n) {
      const int i = tid + n * THREADS;
      const int key = i / C::KPARTS, part = i - key * C::KPARTS;
      if (i < 64 * C::KPARTS && key < klim) *reinterpret_cast<uint4 *>(Kl + key * C::KROW + part * 8) = k[n];
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + n * THREADS;
      const int key = i / C::VPARTS, part = i - key * C::VPARTS;
      if (i < 64 * C::VPARTS && key < vlim) *reinterpret_cast<uint4 *>(Vl + key * C::VROW + part * 8) = v[n];
    }
  }
};

// Global attention, T % 64 == 0: token slots are contiguous in memory, so each thread's chunk addresses are
// fixed up to a per-tile stride -- pointers and LDS offsets are computed once, a tile costs 6 loads + 6 adds.
template <int HD, int THREADS, bool KSWZ = false>
struct StagerLinear {
  using C = Cfg<HD>;
  static constexpr int NK = (64 * C::KPARTS + THREADS - 1) / THREADS;
  static constexpr int NV = (64 * C::VPARTS + THREADS - 1) / THREADS;
  uint4 k[NK], v[NV];
  const u16 *kp[NK], *vp[NV];
  // LDS element offsets from the ring slot's K image.  Surplus lanes (chunk counts are not multiples of the
  // workgroup size) store into the unused 16-byte pad that ends each K row, so every store is unconditional:
  // exec-masked stores would cut the tile loop into extra basic blocks.
  int ko[NK], vo[NV];
  bool kz[NK];                           // chunk lies in the zero padding of the head dim
  static_assert(Cfg<HD>::KROW - Cfg<HD>::HDP == 8, "K rows end in a 16-byte pad");

  __device__ __forceinline__ void init(const AttnParams &p, int b, int head, int tid) {
#pragma unroll
    for (int n = 0; n < NK; ++n) {
      const int i = tid + n * THREADS, ic = min(i, 64 * C::KPARTS - 1);
      const int key = ic / C::KPARTS, part = ic - key * C::KPARTS;
      kz[n] = part * 8 >= HD;
      kp[n] = qkv_at(p, (size_t)b * p.T + key, 1, head) + (kz[n] ? HD - 8 : part * 8);
      ko[n] = i < 64 * C::KPARTS ? key * C::KROW + (KSWZ ? part ^ kswz(key) : part) * 8 : (tid & 63) * C::KROW + C::HDP;
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int i = tid + n * THREADS, ic = min(i, 64 * C::VPARTS - 1);
      const int key = ic / C::VPARTS, part = ic - key * C::VPARTS;
      vp[n] = qkv_at(p, (size_t)b * p.T + key, 2, head) + part * 8;
      vo[n] = i < 64 * C::VPARTS ? 64 * C::KROW + key * C::VROW + part * 8 : (tid & 63) * C::KROW + C::HDP;
    }
  }
  __device__ __forceinline__ void load(size_t tile_stride_elems, int t) {
#pragma unroll
    for (int n = 0; n < NK; ++n) k[n] = *reinterpret_cast<const uint4 *>(kp[n] + tile_stride_elems * t);
#pragma unroll
    for (int n = 0; n < NV; ++n) v[n] = *reinterpret_cast<const uint4 *>(vp[n] + tile_stride_elems * t);
  }
  __device__ __forceinline__ void store(u16 *Kl) const {            // Kl: K image of the ring slot, V image right behind
#pragma unroll
    for (int n = 0; n < NK; ++n) *reinterpret_cast<uint4 *>(Kl + ko[n]) = kz[n] ? make_uint4(0, 0, 0, 0) : k[n];
#pragma unroll
    for (int n = 0; n < NV; ++n) *reinterpret_cast<uint4 *>(Kl + vo[n]) = v[n];
  }
};

template <int HD>
__device__ __forceinline__ void load_q(const AttnParams &p, int b, int wy, int wx, int head, int q0,
                                       bf16x8 (&qf)[Cfg<HD>::KS], int lane) {
  const int g = lane >> 4, c = lane & 15;
  size_t tok;
  const bool valid = token_offset(p, b, wy, wx, q0 + c, tok);
#pragma unroll
  for (int ks = 0; ks < Cfg<HD>::KS; ++ks) {
    union { uint4 u; bf16x8 v; } x;
    x.u = load_chunk<HD>(p, 0, head, valid, tok, ks * 32 + g * 8);
    qf[ks] = x.v;
  }
}

// dst[c*ld + jj] = log2(e) * rel[j0 + sgn*jj] . q_c   for jj in [0, 16*njt).  `rel` is the zero-padded copy
// [rows >= max j + 1][HDP] made by pad_rel_kernel, so every load is unconditional: all njt*KS loads of a table
// are issued back to back (the bounds-checked form compiled to one exec-masked load + vmcnt(0) per MFMA).
template <int HD, int NJT>
__device__ __forceinline__ void build_table(const u16 *rel, int j0, int sgn, const bf16x8 (&qf)[Cfg<HD>::KS],
                                            float *dst, int ld, int lane) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
  union { uint4 u; bf16x8 v; } r[NJT][C::KS];
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    const int j = j0 + sgn * (jt * 16 + c);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
      r[jt][ks].u = *reinterpret_cast<const uint4 *>(rel + (size_t)j * C::HDP + ks * 32 + g * 8);
  }
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) a = S6D_ATTN_MFMA16(r[jt][ks].v, qf[ks], a);
    // C layout: row jj = jt*16 + g*4 + r, col = query c
    float *o = dst + c * ld + jt * 16 + g * 4;
    if ((ld & 3) == 0) {
      *reinterpret_cast<float4 *>(o) = make_float4(a[0] * kLog2e, a[1] * kLog2e, a[2] * kLog2e, a[3] * kLog2e);
    } else {                                              // odd strides (bank spreading): rows are not 16-byte aligned
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = a[r] * kLog2e;
    }
  }
}

// rel (L,HD) -> padded (rows,HDP), zero filled
__global__ void pad_rel_kernel(const u16 *__restrict__ rel_h, const u16 *__restrict__ rel_w, int L, int HD, int HDP,
                               int rows, u16 *__restrict__ out) {
  const int n = rows * HDP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) {
    const int which = i >= n, ii = which ? i - n : i;
    const int j = ii / HDP, d = ii - j * HDP;
    const u16 *src = which ? rel_w : rel_h;
    out[i] = (j < L && d < HD) ? src[j * HD + d] : (u16)0;
  }
}

template <int HD>
__device__ __forceinline__ void store_strip(const AttnParams &p, int b, int wy, int wx, int head, int q0, float l_run,
                                            const f32x4 (&oacc)[Cfg<HD>::DT], int lane) {
  const int g = lane >> 4, c = lane & 15;
  size_t tok;
  if (!token_offset(p, b, wy, wx, q0 + c, tok)) return;
  const float inv = 1.0f / l_run;
  u16 *dst = p.out + tok * (size_t)(p.nh * HD) + head * HD;
#pragma unroll
  for (int dt = 0; dt < Cfg<HD>::DT; ++dt) {
    union { uint2 u; u16 h[4]; } o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o.h[r] = f2bf(oacc[dt][r] * inv);
    *reinterpret_cast<uint2 *>(dst + dt * 16 + g * 4) = o.u;   // O^T rows g*4..g*4+3 = 4 consecutive d
  }
}

// Work item id -> (image, window, head).  All heads of a window read the same token rows (each head a 160-byte run
// of every 7.7 KB row), so they should run on ONE XCD at about the same time: consecutive ids go round-robin over
// the 8 XCDs (observed placement: id % 8), and an XCD's ids walk head-fastest through ITS windows -- the other heads'
// halves of every cache line then hit that XCD's L2 instead of being fetched from HBM once per XCD.
struct WinItem {
  int b, wy, wx, head;
  __device__ __forceinline__ void decode(const AttnParams &p, int id) {
    const int nwin = p.B * p.nwy * p.nwx, cut = (nwin & ~7) * p.nh;
    int win;
    if (id < cut) {
      const int xcd = id & 7, loc = id >> 3;
      head = loc % p.nh;
      win = (loc / p.nh) * 8 + xcd;
    } else {
      head = id % p.nh;
      win = id / p.nh;
    }
    wx = win % p.nwx; win /= p.nwx;
    wy = win % p.nwy; win /= p.nwy;
    b = win;
  }
};

__host__ __device__ inline int win_seq_krows(int T) { return (T + 15) & ~15; }
__host__ __device__ inline int win_seq_vrows(int T) { return (T + 31) & ~31; }
// tile `t` of a sequence against one strip: a full tile, or the tail with the 16-key sub-tiles that exist
template <int HD>
__device__ __forceinline__ void win_seq_tile(const AttnParams &p, const u16 *Kl, const u16 *Vl, int t, int nfull, int tail_subs,
                                             StripState<HD, 1> &st, int lane) {
  using C = Cfg<HD>;
  const float thv[1] = {0.f};
  const u16 *Kt = Kl + (size_t)t * 64 * C::KROW, *Vt = Vl + (size_t)t * 64 * C::VROW;
  if (t < nfull) {
    process_tile<HD, 2, 1>(p, Kt, Vt, t * 64, st, thv, lane);
    return;
  }
  switch (tail_subs) {                                              // wave-uniform
    case 1: process_tile<HD, 2, 1, false, false, 1>(p, Kt, Vt, t * 64, st, thv, lane); break;
    case 2: process_tile<HD, 2, 1, false, false, 2>(p, Kt, Vt, t * 64, st, thv, lane); break;
    case 3: process_tile<HD, 2, 1, false, false, 3>(p, Kt, Vt, t * 64, st, thv, lane); break;
    default: process_tile<HD, 2, 1, false, false, 4>(p, Kt, Vt, t * 64, st, thv, lane); break;
  }
}

// ---- windowed: one workgroup per (image, window, head); every key slot LDS resident ------------------------
template <int HD, int WAVES, bool BIAS>
__global__ __launch_bounds__(WAVES * 64) void attn_window_kernel(AttnParams p) {
  using C = Cfg<HD>;
  constexpr int MODE = BIAS ? 0 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntile = (p.T + 63) / 64;
  // no bias (sequences): COMPACT images -- K rows up to the last 16-key sub-tile that exists, V rows up to the last 32-key step
  // (seq_tile below skips the absent sub-tiles of the tail tile).  257 tokens x head dim 64: 272 x 144 + 288 x 144 = 78.75 KiB instead
  // of 320 rows of both = 95 KiB, i.e. two workgroups per CU: one fetches its item while the other computes (round 4).
  const int krows = BIAS ? ntile * 64 : win_seq_krows(p.T), vrows = BIAS ? ntile * 64 : win_seq_vrows(p.T);
  u16 *Kl = reinterpret_cast<u16 *>(smem);                         // [krows][KROW]
  u16 *Vl = Kl + (size_t)krows * C::KROW;                          // [vrows][VROW]
  float *tabs = reinterpret_cast<float *>(Vl + (size_t)vrows * C::VROW);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float *th = tabs + (size_t)wave * 2 * 16 * p.LT, *tw = th + 16 * p.LT;

  WinItem item;
  item.decode(p, blockIdx.x);
  const int head = item.head, wx = item.wx, wy = item.wy, b = item.b;
  {
    // the loads of up to SB tiles are issued before the first of them is stored: one exposed fetch latency per SB tiles instead of
    // one per tile (T = 257 / 197 keys = 5 / 4 tiles: the whole item in one batch)
    constexpr int SB = 5;
    Stager<HD, WAVES * 64> st[SB];
    for (int t0 = 0; t0 < ntile; t0 += SB) {
#pragma unroll
      for (int i = 0; i < SB; ++i)
        if (t0 + i < ntile) st[i].load(p, b, wy, wx, head, (t0 + i) * 64, tid);
#pragma unroll
      for (int i = 0; i < SB; ++i)
        if (t0 + i < ntile)
          st[i].store(Kl + (size_t)(t0 + i) * 64 * C::KROW, Vl + (size_t)(t0 + i) * 64 * C::VROW, tid, krows - (t0 + i) * 64,
                      vrows - (t0 + i) * 64);
    }
  }
  __syncthreads();

  const int nstrip = (p.T + 15) / 16;
  for (int strip = wave; strip < nstrip; strip += WAVES) {
    const int q0 = strip * 16;
    StripState<HD, 1> st;
    load_q<HD>(p, b, wy, wx, head, q0, st.qf[0], lane);
    st.th[0] = th; st.tw[0] = tw;
    if (BIAS) {
      for (int jt = 0; jt < p.LT / 16; ++jt) {
        build_table<HD, 1>(p.rel_h, jt * 16, 1, st.qf[0], th + jt * 16, p.LT, lane);
        build_table<HD, 1>(p.rel_w, jt * 16, 1, st.qf[0], tw + jt * 16, p.LT, lane);
      }
    }
    const int qi = min(q0 + (lane & 15), p.T - 1);
    st.qy[0] = div_S(p, qi); st.qx[0] = qi - st.qy[0] * p.S;
    st.m_run[0] = -1e30f; st.lacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) st.twr[0][i] = 0.f;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) st.oacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float thv[1] = {0.f};
    if (BIAS) {
      for (int t = 0; t < ntile; ++t)
        process_tile<HD, MODE, 1>(p, Kl + (size_t)t * 64 * C::KROW, Vl + (size_t)t * 64 * C::VROW, t * 64, st, thv, lane);
    } else {
      const int nfull = p.T >> 6, tail_subs = ((p.T & 63) + 15) >> 4;
      for (int t = 0; t < ntile; ++t) win_seq_tile<HD>(p, Kl, Vl, t, nfull, tail_subs, st, lane);
    }
    store_strip<HD>(p, b, wy, wx, head, q0, st.lacc[0][0], st.oacc[0], lane);
  }
}

// One pass of the row-padded window kernel: NS query rows (qy0, qy0 + rstride) against all key rows; the rows share
// every K / V fragment read.  qf: Q fragments of the rows; tabs: NS x [rel_h | rel_w][16][32] floats of LDS.
// A window holds at most 16 key rows, so the whole score strip of a query row (SRC x 4 values per lane) stays in
// registers: one pass of QK^T MFMAs, ONE max / sum exchange per query row, one pass of PV MFMAs -- no running
// max, no rescale of the accumulators, no per-tile cross-lane round trips.  SRC = compile-time key-row count
// (even); EXACT: S == SRC, nothing to mask but the kx >= S columns (carried by twr).
// Per-query bias tables of NR query rows (qy0, qy0 + rstride): rel_h / rel_w dot q, through the wave's own LDS
// scratch `tabs` (NR x [rel_h | rel_w][16][32] floats).  rel_w collapses to 4 registers per row (kx = g*4+r is fixed
// per lane; out-of-window columns carry -1e30 there, which is the column mask).  Depends only on Q and the
// position tables, so the kernel runs it while the K / V staging loads are in flight.
template <int HD>
struct RelFrags {                                                   // MFMA A fragments of the padded position tables, rows 0..31
  union { uint4 u; bf16x8 v; } rh[2][Cfg<HD>::KS], rw[2][Cfg<HD>::KS];
  __device__ __forceinline__ void load(const AttnParams &p, int lane) {
    using C = Cfg<HD>;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        rh[jt][ks].u = *reinterpret_cast<const uint4 *>(p.rel_h + (size_t)(jt * 16 + c) * C::HDP + ks * 32 + g * 8);
        rw[jt][ks].u = *reinterpret_cast<const uint4 *>(p.rel_w + (size_t)(jt * 16 + c) * C::HDP + ks * 32 + g * 8);
      }
  }
};

constexpr int W16_LT = 36;   // scratch table row stride (floats): 144 B rows keep the float4 writes and the per-query
                             // word reads (16 distinct rows per instruction) on distinct banks; 32 was 8-way conflicted

// Bias registers of NR query rows: rel_w / rel_h dot q through the wave's own LDS scratch `tab` ([16][W16_LT]
// floats, reused table after table: same-wave LDS operations execute in order).  Per lane (query c, group g):
//   twr[n][r]  = log2e * rel_w[c - kx + S - 1] . q_c, kx = g*4 + r   (-1e30 for kx >= S: the column mask)
//   thv[n][ky] = log2e * rel_h[qy_n - ky + S - 1] . q_c             (-1e30 for ky >= S: the row mask)
// Depends only on Q and the position tables: the kernel runs it while the K / V staging loads are in flight.
template <int HD, int NR, int SRC>
__device__ __forceinline__ void win16_tables(const RelFrags<HD> &rf, int S, int qy0, int rstride,
                                             const bf16x8 (*qf)[Cfg<HD>::KS], float *tab, float (*twr)[4],
                                             float (*thv)[16], int lane) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int qy = min(qy0 + n * rstride, S - 1);                   // an out-of-window second row is computed, not stored
    f32x4 ah[2], aw[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      ah[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
      aw[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        aw[jt] = S6D_ATTN_MFMA16(rf.rw[jt][ks].v, qf[n][ks], aw[jt]);
        ah[jt] = S6D_ATTN_MFMA16(rf.rh[jt][ks].v, qf[n][ks], ah[jt]);
      }
    }
    // C layout: row jj = jt*16 + g*4 + r, col = query c
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
      *reinterpret_cast<float4 *>(tab + c * W16_LT + jt * 16 + g * 4) =
          make_float4(aw[jt][0] * kLog2e, aw[jt][1] * kLog2e, aw[jt][2] * kLog2e, aw[jt][3] * kLog2e);
    // hipemu: wave rendezvous (same-wave LDS order; on the GPU the wave's LDS operations execute in program order)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kx = g * 4 + r;
      const float v = tab[c * W16_LT + min(max(c - kx + S - 1, 0), 31)];
      twr[n][r] = (kx < S && c < S) ? v : -1e30f;
    }
    // hipemu: wave rendezvous (same-wave LDS order; on the GPU the wave's LDS operations execute in program order)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
      *reinterpret_cast<float4 *>(tab + c * W16_LT + jt * 16 + g * 4) =
          make_float4(ah[jt][0] * kLog2e, ah[jt][1] * kLog2e, ah[jt][2] * kLog2e, ah[jt][3] * kLog2e);
    // hipemu: wave rendezvous (same-wave LDS order; on the GPU the wave's LDS operations execute in program order)
#pragma unroll
    for (int ky = 0; ky < SRC; ++ky) {
      const float v = tab[c * W16_LT + min(max(qy - ky + S - 1, 0), 31)];
      thv[n][ky] = ky < S ? v : -1e30f;
    }
    // hipemu: wave rendezvous (same-wave LDS order; on the GPU the wave's LDS operations execute in program order)
  }
}

// One pass of the row-padded window kernel: NS query rows against all key rows; the rows share every K / V
// fragment read.  A window holds at most 16 key rows, so a query row's whole score strip (SRC x 4 values per
// lane) stays in registers and the softmax is exact two-pass, not online:
//   A  S^T = K Q^T for every key row; s' = scale*acc + twr (one FMA), row maxima join a per-lane running max;
//   B  one cross-lane exchange for the strip maximum;
//   C  per 32-key step: P = exp2(s' - (m - thv)), then O^T += V^T P^T -- the V fragments of the step are requested
//      before the exponentials so their LDS latency hides under them, and the MFMAs of step u run under the
//      exponentials of step u+1.  The row sum comes from the matrix core too (an all-ones A fragment), so it is the
//      sum of the SAME bf16-rounded P that multiplies V and costs no VALU adds or lane exchanges.
// SRC = compile-time key-row count (even); EXACT: S == SRC (no unstaged key rows to skip).
struct Win16NoMid {
  __device__ __forceinline__ void operator()() const {}
};
// KROWT: K image row stride (elements); MID: called half way through the PV pass (the Q fragments are dead since the score pass,
// and by then half of the score registers are free).
template <int HD, int NS, int SRC, bool EXACT, int KROWT = Cfg<HD>::KROW, bool KSWZ = false, class MID = Win16NoMid>
__device__ __forceinline__ void win16_pass(const AttnParams &p, const u16 *Kl, const u16 *Vl, int S, int SR, int b,
                                           int wy, int wx, int head, int qy0, int rstride,
                                           const bf16x8 (&qf)[NS][Cfg<HD>::KS], const float (*twr)[4],
                                           const float (*thv)[16], int lane, MID mid = MID()) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
  const int Cc = p.nh * HD;
  // ---- A: scores of every key row: S^T[key row ky][kx = g*4+r][query c] ----------------------------------
  float sp[NS][SRC][4], m[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) m[n] = -1e29f;                     // finite even if a whole strip is padding
  f32x4 cb[NS];
  {
    const float inv = 1.0f / p.scale_log2;
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[n][r] = twr[n][r] * inv;          // masked columns: -1e30 / scale, still finite
  }
  bf16x8 kf[2][C::KS];
  // compact (LDS-DMA staged) images keep the chunk pairs of rows with (kx >> 2 ^ kx >> 3) & 1 swapped (S6D_GLB_KSWZ above; kx = c here)
  const int gk = KSWZ ? (g ^ kswz(c)) : g;
  auto kload = [&](int ky, bf16x8 (&dst)[C::KS]) {
    const int kyc = EXACT ? ky : min(ky, SR - 1);                   // rows >= SR are not staged: re-read a staged one
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      dst[ks] = *reinterpret_cast<const bf16x8 *>(Kl + (kyc * 16 + c) * KROWT + ks * 32 + gk * 8);
      // compact K rows (KROWT < padded head dim): the k range past HD is not staged -- the lane reads into the next row and
      // the fragment is zeroed here (the matching Q fragment is zero too, but 0 x a stray Inf would not be)
      if (KROWT < C::HDP && ks * 32 + 32 > HD && ks * 32 + g * 8 >= HD) {
        union { bf16x8 v; uint4 u; } z;
        z.u = make_uint4(0, 0, 0, 0);
        dst[ks] = z.v;
      }
    }
  };
  kload(0, kf[0]);
#pragma unroll
  for (int ky = 0; ky < SRC; ++ky) {
    if (ky + 1 < SRC) kload(ky + 1, kf[(ky + 1) & 1]);
    // the column bias rides the matrix core (round 5): the score chain starts from C = tw / scale_log2, so sp holds
    // (q.k + tw / scale) and the scale is applied inside the exponential's fma below -- one VALU instruction per score less
    f32x4 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) acc[n] = cb[n];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int n = 0; n < NS; ++n)
        acc[n] = S6D_ATTN_MFMA16(kf[ky & 1][ks], qf[n][ks], acc[n]);
#pragma unroll
    for (int n = 0; n < NS; ++n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[n][ky][r] = acc[n][r];
      const float mk = __builtin_fmaf(fmaxf(fmaxf(acc[n][0], acc[n][1]), fmaxf(acc[n][2], acc[n][3])), p.scale_log2, thv[n][ky]);
      m[n] = fmaxf(m[n], mk);
    }
  }
  // ---- B: strip maximum ----------------------------------------------------------------------------------------
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    m[n] = fmaxf(m[n], __shfl_xor(m[n], 16));
    m[n] = fmaxf(m[n], __shfl_xor(m[n], 32));
  }
  // ---- C: P and O^T = V^T P^T over 32-key steps (key rows u, u+1) -------------------------------------------
  f32x4 oacc[NS][C::DT], lacc[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    lacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  union { bf16x8 v; u16 hh[8]; } ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones.hh[i] = S6D_ATTN_ONE;          // 1.0 in the element type
#pragma unroll
  for (int u = 0; u < SRC; u += 2) {
    // half of the score registers have been consumed: room for what `mid` brings in (the next item's Q fragments)
    if (u == ((SRC / 2 + 1) & ~1)) {
      __builtin_amdgcn_sched_barrier(0);                              // keep the loads `mid` issues from being hoisted into the first half
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (EXACT || u < SR) {                                          // wave-uniform; K / V rows >= SR are not staged
      union { bf16x8 v; s16x4 q[2]; } va[C::DT];
      const u16 *vrow = Vl + (u * 16 + g * 4 + (c >> 2)) * C::VROW + (c & 3) * 4;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        va[dt].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + dt * 16));
        va[dt].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + 16 * C::VROW + dt * 16));
      }
      union { bf16x8 v; u16 hh[8]; } pb[NS];
#pragma unroll
      for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float nb = thv[n][u + h] - m[n];                    // m >= -1e29 (its initial value): a padded query's all-masked strip gives 0, not inf
          for (int r = 0; r < 4; ++r) pb[n].hh[h * 4 + r] = f2bf(fast_exp2(__builtin_fmaf(sp[n][u + h][r], p.scale_log2, nb)));
        }
#pragma unroll
      for (int n = 0; n < NS; ++n) lacc[n] = S6D_ATTN_MFMA16(ones.v, pb[n].v, lacc[n]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int n = 0; n < NS; ++n)
          oacc[n][dt] = S6D_ATTN_MFMA16(va[dt].v, pb[n].v, oacc[n][dt]);
    }
  }
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const int qy = qy0 + n * rstride;
    const int y = wy * p.ws + qy, x = wx * p.ws + c;
    if ((qy < S) && c < S && (y < p.H) && (x < p.W)) {
      const float inv = 1.0f / lacc[n][0];                          // every row of the ones-product is the strip's row sum
      u16 *dst = p.out + ((size_t)(b * p.H + y) * p.W + x) * (size_t)Cc + head * HD;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        union { uint2 u2; u16 hh[4]; } o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o.hh[r] = f2bf(oacc[n][dt][r] * inv);
        *reinterpret_cast<uint2 *>(dst + dt * 16 + g * 4) = o.u2;
      }
    }
  }
}

// ---- round 5: the window pass as ONE stream, without the exact maximum -----------------------------------------------------
// win16_pass is exact two-pass softmax: all SRC x 4 scores of a strip stay in registers (112 for two strips of a 14 x 14 window)
// while the strip maximum is found, then the exponentials and P V run.  The maximum costs 1.25 vector instructions per score, the
// scale + bias another one, and the 112 registers put the kernel at the 256-VGPR limit.  As in process_tile_nomax (global
// attention): P only has to stay representable, bf16 has float32's exponent range, the accumulators are float32.  Here the reference
// value m is the maximum over the FIRST TWO key rows (32 of the 196 keys), known before the first exponential; every 32-key step then
// is score instructions -> exp2(fma(acc, scale_log2, th - m)) -> P V, nothing kept but the accumulators: 2.5 vector instructions
// per score instead of 4.75, no score array (182 VGPRs instead of 236).
// Safety net: a valid row whose sum is not < kNoMaxSumLimit = 2^100 (inf / NaN included) had scores more than ~92 log2 units above its reference
// (or its sum P V left float32: kNoMaxFinite); the
// WAVE then repeats the pass with that row's m raised by 96 (the sum drops by 2^96: any m within ~100 of the true maximum is as good
// as the maximum) until no row is out of range -- K / V stay resident in LDS for the whole item, so the retry is local to the wave,
// it is the same code (no second pass inlined: an exact-pass fallback pushed the kernel into scratch), and it terminates: scores
// of bf16 operands are finite, 40 rounds cover 3840 log2 units; NaN inputs give NaN outputs as they do in the exact pass.
// The column bias rides the matrix core (C = tw / scale_log2), masked columns carry -1e30 / scale there, a padded query's all-masked
// strip is clamped at m >= -1e29 so that its (discarded) P is 0, not inf.
// `next_q` is called once the last score instructions of the (first) round are issued: the caller loads the NEXT item's Q fragments
// into the registers of `qf`; `this_q` reloads THIS item's (before a retry round); after a retry `next_q` is called again at the end.
template <int HD, int SRC, int KROWT, bool KSWZ, class NEXTQ, class THISQ>
__device__ __forceinline__ void win16_pass_stream(const AttnParams &p, const u16 *Kl, const u16 *Vl, int S, int b, int wy, int wx,
                                                  int head, int qy0, int rstride, const bf16x8 (&qf)[2][Cfg<HD>::KS],
                                                  const float (*twr)[4], const float (*thv)[16], int lane, NEXTQ next_q, THISQ this_q) {
  using C = Cfg<HD>;
  constexpr int NS = 2;
  static_assert(SRC % 2 == 0 && SRC >= 4, "whole 32-key steps");
  const int g = lane >> 4, c = lane & 15;
  const int Cc = p.nh * HD;
  const int gk = KSWZ ? (g ^ kswz(c)) : g;
  f32x4 cb[NS];
  {
    const float inv = 1.0f / p.scale_log2;
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[n][r] = twr[n][r] * inv;
  }
  bf16x8 kf[2][C::KS];
  auto kload = [&](int ky, bf16x8 (&dst)[C::KS]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      dst[ks] = *reinterpret_cast<const bf16x8 *>(Kl + (ky * 16 + c) * KROWT + ks * 32 + gk * 8);
      if (KROWT < C::HDP && ks * 32 + 32 > HD && ks * 32 + g * 8 >= HD) {      // compact rows: the k range past HD reads into the next row
        union { bf16x8 v; uint4 u; } z;
        z.u = make_uint4(0, 0, 0, 0);
        dst[ks] = z.v;
      }
    }
  };
  auto scores = [&](const bf16x8 (&k)[C::KS], f32x4 (&acc)[NS]) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NS; ++n) acc[n] = cb[n];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int n = 0; n < NS; ++n) acc[n] = S6D_ATTN_MFMA16(k[ks], qf[n][ks], acc[n]);
  };
  f32x4 s0[NS], s1[NS];
  kload(0, kf[0]);
  kload(1, kf[1]);
  scores(kf[0], s0);
  scores(kf[1], s1);
  float m[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const float m0 = __builtin_fmaf(fmaxf(fmaxf(s0[n][0], s0[n][1]), fmaxf(s0[n][2], s0[n][3])), p.scale_log2, thv[n][0]);
    const float m1 = __builtin_fmaf(fmaxf(fmaxf(s1[n][0], s1[n][1]), fmaxf(s1[n][2], s1[n][3])), p.scale_log2, thv[n][1]);
    float v = fmaxf(fmaxf(m0, m1), -1e29f);
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    m[n] = v;
  }
  union { bf16x8 v; u16 hh[8]; } ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones.hh[i] = S6D_ATTN_ONE;
  f32x4 oacc[NS][C::DT], lacc[NS];
  int round = 0;                                                      // wave-uniform
  for (;;) {
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      lacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) oacc[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < SRC; u += 2) {
      if (u > 0 || round > 0) {                                       // (round 0 enters with the first two rows' scores in s0 / s1)
        kload(u, kf[0]);
        kload(u + 1, kf[1]);
        scores(kf[0], s0);
        scores(kf[1], s1);
      }
      if (u == SRC - 2 && round == 0) {
        // the last score instructions are issued: the Q fragments are dead and the next item's go straight into their registers; the
        // last step's exponentials and P V, the end-of-item wait and the barrier cover the fetch
        __builtin_amdgcn_sched_barrier(0);
        next_q();
        __builtin_amdgcn_sched_barrier(0);
      }
      union { bf16x8 v; s16x4 q[2]; } va[C::DT];
      const u16 *vrow = Vl + (u * 16 + g * 4 + (c >> 2)) * C::VROW + (c & 3) * 4;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        va[dt].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + dt * 16));
        va[dt].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((S6D_LDS(s16x4) *)(vrow + 16 * C::VROW + dt * 16));
      }
      union { bf16x8 v; u16 hh[8]; } pb[NS];
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        const float nb0 = thv[n][u] - m[n], nb1 = thv[n][u + 1] - m[n];
        // (scalar FMAs on purpose: the packed form needs aligned register pairs and put this kernel into scratch)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pb[n].hh[r] = f2bf(fast_exp2(__builtin_fmaf(s0[n][r], p.scale_log2, nb0)));
          pb[n].hh[4 + r] = f2bf(fast_exp2(__builtin_fmaf(s1[n][r], p.scale_log2, nb1)));
        }
      }
#pragma unroll
      for (int n = 0; n < NS; ++n) lacc[n] = S6D_ATTN_MFMA16(ones.v, pb[n].v, lacc[n]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int n = 0; n < NS; ++n) oacc[n][dt] = S6D_ATTN_MFMA16(va[dt].v, pb[n].v, oacc[n][dt]);
    }
    bool any_bad = false;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int qy = qy0 + n * rstride;
      int ovf = 0;                                                    // this lane's slice of the query's O^T column (4 g-lanes per query)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ovf |= !(fabsf(oacc[n][dt][r]) < kNoMaxFinite);
      ovf |= __shfl_xor(ovf, 16);
      ovf |= __shfl_xor(ovf, 32);
      const bool bad = (qy < S) && (c < S) && (!(lacc[n][0] < kNoMaxSumLimit) || ovf);     // a valid query's sum is >= 1 (its reference key)
      m[n] = bad ? m[n] + 96.0f : m[n];
      any_bad |= bad;
    }
    if (!__any(any_bad) || round >= 40) break;
    if (round == 0) this_q();                                         // this item's Q fragments again (next_q() replaced them)
    ++round;
  }
  if (round > 0) next_q();
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const int qy = qy0 + n * rstride;
    const int y = wy * p.ws + qy, x = wx * p.ws + c;
    if ((qy < S) && c < S && (y < p.H) && (x < p.W)) {
      const float inv = 1.0f / lacc[n][0];
      u16 *dst = p.out + ((size_t)(b * p.H + y) * p.W + x) * (size_t)Cc + head * HD;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        union { uint2 u2; u16 hh[4]; } o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o.hh[r] = f2bf(oacc[n][dt][r] * inv);
        *reinterpret_cast<uint2 *>(dst + dt * 16 + g * 4) = o.u2;
      }
    }
  }
}

// ---- windowed, row-padded (S <= 16): key slot = ky*16 + kx, so every 16-key MFMA sub-tile is ONE key row and
// every 16-query strip is ONE query row: the decomposed bias costs one LDS word per sub-tile (rel_h) plus four
// registers (rel_w, kx = g*4+r fixed per lane; out-of-window columns carry -1e30 there, which is the mask).
template <int HD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_window16_kernel(AttnParams p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = p.S, SR = (S + 1) & ~1;                            // key rows, rounded up to a 32-key k-step
  u16 *Kl = reinterpret_cast<u16 *>(smem);                         // [SR*16][KROW]
  u16 *Vl = Kl + (size_t)SR * 16 * C::KROW;                        // [SR*16][VROW]
  float *tabs = reinterpret_cast<float *>(Vl + (size_t)SR * 16 * C::VROW);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  float *tab = tabs + (size_t)wave * 16 * W16_LT;                 // one [16][W16_LT] scratch table per wave

  WinItem item;
  item.decode(p, blockIdx.x);
  const int head = item.head, wx = item.wx, wy = item.wy, b = item.b;
  const int Cc = p.nh * HD;
  // Q fragments of every query row this wave owns are fetched first so their latency hides under the staging
  constexpr int MAXROWS = 2;                                      // rows (wave, wave + WAVES); WAVES >= 8 covers S <= 16
  static_assert(2 * WAVES >= 16, "two rows per wave must cover 16 query rows");
  bf16x8 qfa[MAXROWS][C::KS];
#pragma unroll
  for (int i = 0; i < MAXROWS; ++i) {
    const int qy = wave + i * WAVES;
    const int y = wy * p.ws + qy, x = wx * p.ws + c;
    const bool qwin = c < S && qy < S, qimg = qwin && (y < p.H) && (x < p.W);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      union { uint4 u; bf16x8 v; } t;
      const int d0 = ks * 32 + g * 8;
      const int dc = d0 < HD ? d0 : HD - 8;
      const size_t tokc = (size_t)(b * p.H + min(y, p.H - 1)) * p.W + min(x, p.W - 1);
      const u16 *src = qimg ? qkv_at(p, tokc, 0, head) + dc : p.qkv_bias + head * HD + dc;
      t.u = *reinterpret_cast<const uint4 *>(src);
      if (!(qwin && d0 < HD)) t.u = make_uint4(0, 0, 0, 0);
      qfa[i][ks] = t.v;
    }
  }
  RelFrags<HD> rf;
  rf.load(p, lane);
  // stage K and V of the whole window (bias vector for out-of-image slots, zeros for padding slots): UN
  // independent 16-byte loads per thread are in flight before any is written to LDS, and the first batch
  // flies under the bias-table MFMAs (which need only Q and the position tables).
  constexpr int UN = 10;
  const int total = SR * 16 * (C::KPARTS + C::VPARTS);
  uint4 sv[UN];
  u16 *sdst[UN];
  auto issue = [&](int i0) {
#pragma unroll
    for (int n = 0; n < UN; ++n) {
      const int i = i0 + n * WAVES * 64;
      const int iq = min(i, total - 1);
      const bool isv = iq >= SR * 16 * C::KPARTS;
      const int ii = isv ? iq - SR * 16 * C::KPARTS : iq;
      const int parts = isv ? C::VPARTS : C::KPARTS;
      const int slot = ii / parts, part = ii - slot * parts;
      const int ky = slot >> 4, kx = slot & 15;
      const int y = wy * p.ws + ky, x = wx * p.ws + kx;
      sdst[n] = (i < total) ? (isv ? Vl + slot * C::VROW + part * 8 : Kl + slot * C::KROW + part * 8) : nullptr;
      const bool img = (y < p.H) && (x < p.W);
      const int dc = part * 8 < HD ? part * 8 : HD - 8;
      const size_t tokc = (size_t)(b * p.H + min(y, p.H - 1)) * p.W + min(x, p.W - 1);
      const int sel = (isv ? 2 : 1) * Cc + head * HD + dc;
      const u16 *src = img ? qkv_at(p, tokc, isv ? 2 : 1, head) + dc : p.qkv_bias + sel;
      sv[n] = *reinterpret_cast<const uint4 *>(src);
      if (!(ky < S && kx < S && part * 8 < HD)) sv[n] = make_uint4(0, 0, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int n = 0; n < UN; ++n)
      if (sdst[n]) *reinterpret_cast<uint4 *>(sdst[n]) = sv[n];
  };
  const bool stage = !(kAbl & 1);
  if (stage) issue(tid);
  // bias registers; tables are instantiated for the key-row count the pass will use
  float twr[MAXROWS][4], thv[MAXROWS][16];
  if (!(kAbl & 2)) {
    if (S > WAVES) {
      if (S == 14)
        win16_tables<HD, 2, 14>(rf, S, wave, WAVES, qfa, tab, twr, thv, lane);
      else
        win16_tables<HD, 2, 16>(rf, S, wave, WAVES, qfa, tab, twr, thv, lane);
    } else if (wave < S) {
      win16_tables<HD, 1, 8>(rf, S, wave, WAVES, qfa, tab, twr, thv, lane);
    }
  }
  if (stage) {
    commit();
    for (int i0 = tid + UN * WAVES * 64; i0 < total; i0 += UN * WAVES * 64) {
      issue(i0);
      commit();
    }
  }
  __syncthreads();

  // S <= WAVES: one query row per wave; otherwise two rows per wave share every K / V fragment read:
  // rows (wave, wave + WAVES).
  if (S <= WAVES) {
    if (wave < S && !(kAbl & 2)) {
      const bf16x8 (&q1)[1][C::KS] = reinterpret_cast<const bf16x8 (&)[1][C::KS]>(qfa[0]);
      win16_pass<HD, 1, 8, false>(p, Kl, Vl, S, SR, b, wy, wx, head, wave, WAVES, q1, twr, thv, lane);
    }
  } else if (!(kAbl & 2)) {
    static_assert(MAXROWS >= 2 || WAVES >= 16, "row bookkeeping");
    const bf16x8 (&q2)[2][C::KS] = reinterpret_cast<const bf16x8 (&)[2][C::KS]>(qfa[0]);
    if (S == 14) {
      win16_pass<HD, 2, 14, true>(p, Kl, Vl, S, SR, b, wy, wx, head, wave, WAVES, q2, twr, thv, lane);
    } else
      win16_pass<HD, 2, 16, false>(p, Kl, Vl, S, SR, b, wy, wx, head, wave, WAVES, q2, twr, thv, lane);
  }
}

// ---- windowed, row-padded, PERSISTENT: one 8-wave workgroup per CU walks its share of the (window, head) items; the K / V
// images of item i + 1 arrive by LDS-DMA into the second half of LDS while item i is computed from the first.
//
// Why: the one-item-per-workgroup kernel above spends an item as [fetch 107 KB | arithmetic | store]: with one 98-KiB workgroup
// per CU nothing overlaps, and the CU's fetch path (~10 B/clk/CU from HBM, MI355X_MICROARCH.md) is idle for half of the item.
// Two smaller workgroups per CU did not help (each needs its own copy of the window: 0.375 vs 0.358 ms per 16 frames); keeping
// the fetch path busy does -- a register-free prefetch of the NEXT item, which LDS-DMA provides.
//   * LDS (all 160 KiB): 2 x [K image 224 slots x 11 chunks (10 data + 1 pad: 176-B rows are bank-conflict free for the fragment
//     reads) padded to 39 KiB | V image 224 x 10 chunks = 35 KiB] + the two padded rel-pos tables (12 KiB, read as A fragments per
//     item instead of being held in 48 VGPRs across the loop).  The per-wave scratch tables of win16_tables live in the V area
//     of the buffer that is being FILLED: the K image of the next item goes out first, the V image after the tables are done.
//   * an image is lane-linear for the DMA (64 consecutive 16-byte chunks per instruction); the lane works out which (slot, part)
//     its chunk is: in-window in-image -> the token's k / v run, out-of-image -> the qkv bias (quirk Q2 of SURVEY: padded tokens
//     carry the bias), outside the 14 x 14 window or the pad chunk -> a 16-byte zero in global memory.
//   * Q fragments of item i + 1 are loaded into the registers of item i's Q fragments half way through item i's PV pass (they are
//     dead since the score pass, and half of the score registers are free by then).
// Measured (16 frames, 6400 items): 0.305 ms against 0.350 ms for the one-item kernel (profiles/r02_win16_variants.txt).  Tried on
// top and dropped, both slower: keeping the packed output rows in registers and storing them after the end-of-item wait (so that the
// wait does not cover the stores): 0.396 ms; that plus folding the row bias into the scores to free 28 registers in the PV pass:
// 0.431 ms (the kernel sits at the 256-VGPR limit: every extra live value in the PV pass became scratch traffic in the hot loop).
__device__ const uint4 g_win16_zero = {0u, 0u, 0u, 0u};

#ifdef HIPEMU
#define S6D_ATTN_VMCNT0() hipemu::vmcnt_wait(0)
#else
#define S6D_ATTN_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#define S6D_ATTN_GLOBAL(T) __attribute__((address_space(1))) T

// One 1-KiB DMA piece: lane l's 16 bytes at src -> LDS byte address lds + 16 l (lds wave-uniform).  Issued as inline asm, not through
// __builtin_amdgcn_global_load_lds: with the builtin in the loop hipcc puts an s_waitcnt vmcnt(0) in front of the first
// ds_read_b64_tr_b16 of every tile (it cannot tell that the reads touch another ring slot), which drains the two-tile look-ahead;
// the waits here are the counted ones written out below.  M0 is on the clobber list (the library is built with -Wno-inline-asm: hipcc
// warns about reserved registers there); nothing else in the kernel uses it.
#ifdef HIPEMU
#define S6D_ATTN_DMA16(src, lds_ptr) __builtin_amdgcn_global_load_lds((const S6D_ATTN_GLOBAL(void) *)(src), (lds_ptr), 16, 0, 0)
#else
__device__ __forceinline__ void attn_dma16(const void *src, S6D_LDS(char) *dst) {
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);
  // s_nop: one wait state between the SALU write of M0 and the LDS-DMA that reads it (hipcc puts the same nop after its own s_mov m0)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(a) : "memory", "m0");
}
#define S6D_ATTN_DMA16(src, lds_ptr) attn_dma16((src), (lds_ptr))
#endif


// S14: the kernel is instantiated once for 14 x 14 windows (the ViT-H configuration: every key row exists, win16_pass<.., 14, true>)
// and once for the other sizes (win16_pass<.., 16, false>).  As ONE kernel holding both passes it needed 256 VGPRs + 21 spilled
// registers, stored to scratch for every item on either path; the 14 x 14 instantiation alone takes 239 VGPRs and no scratch
// (round 4: tools/kernel_resources.py).
template <int HD, bool S14>
__global__ __launch_bounds__(512) void attn_window16p_kernel(AttnParams p, int nitems) {
  using C = Cfg<HD>;
  constexpr int WAVES = 8;
  constexpr int KROWT = HD + 8, KCH = KROWT / 8, VCH = C::VROW / 8;   // chunks (16 B) per K / V row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = p.S, SR = (S + 1) & ~1;
  const int kchunks = SR * 16 * KCH, vchunks = SR * 16 * VCH;
  const int kinstr = (kchunks + 63) >> 6, vinstr = (vchunks + 63) >> 6;
  const int kbytes = kinstr << 10, bufbytes = kbytes + (vinstr << 10);
  u16 *relh = reinterpret_cast<u16 *>(smem + 2 * bufbytes);        // [32][HDP] padded rel_pos_h, then rel_pos_w
  u16 *relw = relh + 32 * C::HDP;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int Cc = p.nh * HD;
  // rel tables -> LDS, once.  A row is HDP / 8 = 12 (8) chunks of 16 bytes, so the 16 rows of a fragment read start on only 4 (2) of
  // the 16 bank slots: every one of the 12 fragment reads per item was a 4-way conflict -- 2.3 k of a workgroup's ~28 k cycles per
  // item, and the whole of the kernel's SQ_LDS_BANK_CONFLICT count (14.5 M per launch = 6400 items x 2.3 k, profiles/r04_sq_summary
  // .json).  There is no LDS left for padded rows (2 x 74 KiB images + 12 KiB tables = 160 KiB), so the chunks of
  // row r are stored at chunk ^ ((r >> 2) & 3): inside a group of four chunks, 16 rows x one chunk index -> 16 distinct slots.
  constexpr int RCH = C::HDP / 8;
  for (int i = tid; i < 2 * 32 * RCH; i += 512) {
    const bool w = i >= 32 * RCH;
    const int ii = w ? i - 32 * RCH : i;
    const int row = ii / RCH, ch = ii - row * RCH;
    reinterpret_cast<uint4 *>(w ? relw : relh)[row * RCH + (ch ^ ((row >> 2) & 3))] = reinterpret_cast<const uint4 *>(w ? p.rel_w : p.rel_h)[ii];
  }

  // ---- DMA of one image: `which` 1 = K (row KCH chunks, the last one padding), 2 = V.
  // Which chunk of the image a lane feeds in the wave's i-th DMA instruction -- key row, key column, head-dim part, inside the
  // S x S window or not -- does not depend on the item: it is computed ONCE per kernel and kept packed in one register per
  // instruction slot (round 5).  Until then every instruction recomputed it (a division by the row's chunk count, the swizzle, two
  // 64-bit token products: ~56 instructions per DMA instruction, ~10 per wave and item: a quarter of the wave's instruction
  // stream).  Per item only the window origin, the image test at the frame's edge and one 64-bit multiply-add remain.
  constexpr int MAXI = S14 ? 5 : 6;                                  // DMA instructions per wave and image: ceil(14 (16) x 16 x 11 / 64 / 8)
  // the slots are unrolled: an image with more 1-KiB pieces than WAVES x MAXI would silently lose its tail -- the
  // largest window this instantiation serves (14 rows, or 16 = the launcher's limit) must fit for both images
  static_assert(((S14 ? 14 : 16) * 16 * (KCH > VCH ? KCH : VCH) + 63) / 64 <= WAVES * MAXI,
                "attn_window16p_kernel: K / V image has more DMA pieces than the unrolled slots (raise MAXI)");
  unsigned kvc[MAXI];                     // per instruction slot: K in bits 0-12, V in bits 16-28: ky | kx << 4 | part << 8 | inwin << 12
  {
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      unsigned both = 0;
#pragma unroll
      for (int which = 1; which <= 2; ++which) {
        const int rowch = which == 1 ? KCH : VCH, nch = which == 1 ? kchunks : vchunks;
        const int k = wave + i * WAVES;
        const int j = min((k << 6) + lane, nch - 1);                  // chunk of the image (tail lanes repeat the last chunk's source)
        const int slot = j / rowch, pp = j - slot * rowch;
        const int ky = slot >> 4, kx = slot & 15;
        const int part = (S6D_WIN16_KSWZ && which == 1 && pp * 8 < HD) ? pp ^ kswz(kx) : pp;     // K: chunk pairs swapped on the source side
        const bool inwin = ky < S && kx < S && part * 8 < HD;
        const unsigned v = (unsigned)(ky & 15) | ((unsigned)kx << 4) | ((unsigned)part << 8) | ((unsigned)inwin << 12);
        both |= which == 1 ? v : v << 16;
      }
      kvc[i] = both;
    }
  }
  auto stage_image = [&](const WinItem &it, int which, char *dst_base) __attribute__((always_inline)) {
    const int ninstr = which == 1 ? kinstr : vinstr;
    const int y0 = it.wy * p.ws, x0 = it.wx * p.ws, bH = it.b * p.H;
    const u16 *base = qkv_at(p, 0, which, it.head);                   // wave-uniform
    const u16 *bias_base = p.qkv_bias + which * Cc + it.head * HD;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int k = wave + i * WAVES;
      if (k < ninstr) {                                               // wave-uniform
        unsigned cc = which == 1 ? kvc[i] : kvc[i] >> 16;
#ifndef HIPEMU
        asm volatile("" : "+v"(cc));        // opaque per item: the unpacked fields must not be hoisted out of the item loop (40 registers)
#endif
        const int ky = (int)(cc & 15u), kx = (int)((cc >> 4) & 15u), po = min((int)((cc >> 8) & 15u) * 8, HD - 8);
        const int y = y0 + ky, x = x0 + kx;
        const bool img = (y < p.H) && (x < p.W);
        const unsigned tok = (unsigned)(bH + min(y, p.H - 1)) * (unsigned)p.W + (unsigned)min(x, p.W - 1);
        const u16 *s_img = base + (size_t)tok * (size_t)p.tok_stride + po;
        const u16 *src = img ? s_img : bias_base + po;
        const void *sp = (cc >> 12) & 1u ? (const void *)src : (const void *)&g_win16_zero;
        S6D_LDS(char) *dst = (S6D_LDS(char) *)dst_base + (k << 10);
#if S6D_WIN16_ASM_DMA
        S6D_ATTN_DMA16(sp, dst);
#else
        __builtin_amdgcn_global_load_lds((const S6D_ATTN_GLOBAL(void) *)sp, dst, 16, 0, 0);
#endif
      }
    }
  };
  constexpr int MAXROWS = 2;
  bf16x8 qfa[MAXROWS][C::KS];
  // RAW: loads only (the zero mask of the out-of-window / padded-head-dim chunks is lane-constant and applied by mask_q at the
  // start of the item that uses the fragments, behind the end-of-item wait)
  auto load_q = [&](const WinItem &it, bool raw, bf16x8 (&qdst)[MAXROWS][C::KS]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MAXROWS; ++i) {
      const int qy = wave + i * WAVES;
      const int y = it.wy * p.ws + qy, x = it.wx * p.ws + c;
      const bool qwin = c < S && qy < S, qimg = qwin && (y < p.H) && (x < p.W);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        union { uint4 u; bf16x8 v; } t;
        const int d0 = ks * 32 + g * 8;
        const int dc = d0 < HD ? d0 : HD - 8;
        const size_t tokc = (size_t)(it.b * p.H + min(y, p.H - 1)) * p.W + min(x, p.W - 1);
        const u16 *src = qimg ? qkv_at(p, tokc, 0, it.head) + dc : p.qkv_bias + it.head * HD + dc;
        t.u = *reinterpret_cast<const uint4 *>(src);
        if (!raw && !(qwin && d0 < HD)) t.u = make_uint4(0, 0, 0, 0);
        qdst[i][ks] = t.v;
      }
    }
  };
  auto mask_q = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MAXROWS; ++i) {
      const bool qwin = c < S && wave + i * WAVES < S;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        union { uint4 u; bf16x8 v; } t;
        t.v = qfa[i][ks];
        if (!(qwin && ks * 32 + g * 8 < HD)) t.u = make_uint4(0, 0, 0, 0);
        qfa[i][ks] = t.v;
      }
    }
  };

  WinItem cur, nxt;
  int id = blockIdx.x;
  cur.decode(p, id);
  load_q(cur, false, qfa);
  stage_image(cur, 1, smem);
  stage_image(cur, 2, smem + kbytes);
  S6D_ATTN_VMCNT0();
  __syncthreads();
  int buf = 0;
  long long wtk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wsum[6] = {0, 0, 0, 0, 0, 0};   // S6D_G64_TIMING probe build: phase clocks, see below
  for (; id < nitems; id += gridDim.x) {
    char *cb = smem + buf * bufbytes, *nb = smem + (buf ^ 1) * bufbytes;
    const bool more = id + (int)gridDim.x < nitems;
    S6D_TICK(wtk, 0);
    if (more) {
      nxt.decode(p, id + gridDim.x);
      stage_image(nxt, 1, nb);                                       // K image of the next item; its V area is this item's scratch
    }
    // bias registers of this item (rel-pos tables as A fragments from LDS; scratch tables in the V area of the other buffer)
    float twr[MAXROWS][4], thv[MAXROWS][16];
    {
      RelFrags<HD> rf;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          const int ro = (jt * 16 + c) * C::HDP + ((ks * 4 + g) ^ (((jt * 16 + c) >> 2) & 3)) * 8;     // chunk swizzle of the copy above
          rf.rh[jt][ks].u = *reinterpret_cast<const uint4 *>(relh + ro);
          rf.rw[jt][ks].u = *reinterpret_cast<const uint4 *>(relw + ro);
        }
      float *tab = reinterpret_cast<float *>(nb + kbytes) + (size_t)wave * 16 * W16_LT;
      if (S == 14)
        win16_tables<HD, 2, 14>(rf, S, wave, WAVES, qfa, tab, twr, thv, lane);
      else
        win16_tables<HD, 2, 16>(rf, S, wave, WAVES, qfa, tab, twr, thv, lane);
    }
    S6D_TICK(wtk, 1);
    __builtin_amdgcn_s_barrier();                                    // every wave is done with its scratch table (raw: the K DMA stays in flight)
    S6D_TICK(wtk, 2);
    if (more) stage_image(nxt, 2, nb + kbytes);
    S6D_TICK(wtk, 3);
    const u16 *Kl = reinterpret_cast<const u16 *>(cb), *Vl = reinterpret_cast<const u16 *>(cb + kbytes);
    const bf16x8 (&q2)[2][C::KS] = reinterpret_cast<const bf16x8 (&)[2][C::KS]>(qfa[0]);
    if (S14 && S6D_WIN16_STREAM) {
      // the streaming pass keeps this item's Q fragments live to its last score instructions; the next item's are requested right
      // there, into the same registers (a retry round of the pass fetches this item's again)
      auto next_q = [&]() __attribute__((always_inline)) {
        if (more) load_q(nxt, S6D_WIN16_QDEFER != 0, qfa);
      };
      auto this_q = [&]() __attribute__((always_inline)) { load_q(cur, false, qfa); };
      win16_pass_stream<HD, 14, KROWT, S6D_WIN16_KSWZ != 0>(p, Kl, Vl, S, cur.b, cur.wy, cur.wx, cur.head, wave, WAVES, q2, twr, thv, lane,
                                                            next_q, this_q);
    } else {
      auto mid = [&]() __attribute__((always_inline)) {              // half way through the PV pass: this item's Q fragments are long dead
        if (more) load_q(nxt, S6D_WIN16_QDEFER != 0, qfa);
      };
      if (S14)
        win16_pass<HD, 2, 14, true, KROWT, S6D_WIN16_KSWZ != 0>(p, Kl, Vl, S, SR, cur.b, cur.wy, cur.wx, cur.head, wave, WAVES, q2, twr, thv, lane, mid);
      else
        win16_pass<HD, 2, 16, false, KROWT, S6D_WIN16_KSWZ != 0>(p, Kl, Vl, S, SR, cur.b, cur.wy, cur.wx, cur.head, wave, WAVES, q2, twr, thv, lane, mid);
    }
    S6D_TICK(wtk, 4);
    S6D_ATTN_VMCNT0();                                               // next item's images and Q have landed (and this item's stores)
    S6D_TICK(wtk, 5);
    __syncthreads();
    S6D_TICK(wtk, 6);
    if (S6D_G64_TIMING)
#pragma unroll
      for (int i = 0; i < 6; ++i) wsum[i] += wtk[i + 1] - wtk[i];
    // unconditional: on the last item the fragments are stale and unused, but hipcc's wait-count model then sees every path into the
    // loop header with no load in flight (otherwise it waits vmcnt(5..0) at the fragments' first use, i.e. for the K DMA issued before it)
    if (S6D_WIN16_QDEFER) mask_q();
    cur = nxt;
    buf ^= 1;
  }
  // probe build: [K DMA issue + bias tables | barrier | V DMA issue | score / softmax / P V passes | vmcnt(0) | barrier] per wave of
  // workgroup 8, 320 bytes behind the output tensor (tools/attn_time.py)
  if (S6D_G64_TIMING && blockIdx.x == 8 && lane == 0) {
    long long *dbg = reinterpret_cast<long long *>(p.out + (size_t)p.B * p.H * p.W * p.nh * HD);
#pragma unroll
    for (int i = 0; i < 6; ++i) dbg[wave * 6 + i] = wsum[i];
  }
}

// ---- global: one workgroup per (image, head, 128-query tile); KV tiles stream through a 2-deep LDS ring -----
// Each wave owns NS = 2 strips (32 queries): K/V fragments and every staged tile are shared by twice the math.
template <int HD, int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void attn_global_kernel(AttnParams p) {
  using C = Cfg<HD>;
  constexpr int NS = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KBYTES = 64 * C::KROW * 2, VBYTES = 64 * C::VROW * 2;
  // ring slot r: K image at r*(KBYTES+VBYTES), V image right behind it
  auto Kbuf = [&](int r) { return reinterpret_cast<u16 *>(smem + r * (KBYTES + VBYTES)); };
  auto Vbuf = [&](int r) { return reinterpret_cast<u16 *>(smem + r * (KBYTES + VBYTES) + KBYTES); };
  float *tabs = reinterpret_cast<float *>(smem + 2 * (KBYTES + VBYTES));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;

  // Workgroup -> (image, head, query tile).  All query tiles of one (image, head) re-read the same 1.3 MB of
  // K/V: keep them on ONE XCD (observed placement: block id % 8) so the re-reads hit that XCD's 4 MB L2
  // instead of streaming from HBM once per query tile.  Pure speed choice; any placement is correct.
  const int nqt = (p.T + WAVES * 16 * NS - 1) / (WAVES * 16 * NS);
  int id = blockIdx.x;
  const int nbh = p.B * p.nh;
  int qt, bh;
  if ((nbh & 7) == 0) {
    const int xcd = id & 7, loc = id >> 3;
    qt = loc % nqt;
    bh = (loc / nqt) * 8 + xcd;
  } else {
    qt = id % nqt;
    bh = id / nqt;
  }
  const int head = bh % p.nh, b = bh / p.nh;
  StripState<HD, NS> st;
  int q0[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    q0[n] = ((qt * WAVES + wave) * NS + n) * 16;
    load_q<HD>(p, b, 0, 0, head, q0[n], st.qf[n], lane);
    const int qi = min(q0[n] + c, p.T - 1);
    st.qy[n] = div_S(p, qi);
    st.qx[n] = qi - st.qy[n] * p.S;
    st.m_run[n] = -1e30f;
    st.lacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) st.twr[n][i] = 0.f;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) st.oacc[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.th[n] = nullptr;
    st.tw[n] = nullptr;
  }
  float *thm[NS] = {nullptr, nullptr};
  if (MODE == 0) {
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      float *th = tabs + (size_t)(wave * NS + n) * 2 * 16 * p.LT, *tw = th + 16 * p.LT;
      for (int jt = 0; jt < p.LT / 16; ++jt) {
        build_table<HD, 1>(p.rel_h, jt * 16, 1, st.qf[n], th + jt * 16, p.LT, lane);
        build_table<HD, 1>(p.rel_w, jt * 16, 1, st.qf[n], tw + jt * 16, p.LT, lane);
      }
      st.th[n] = th;
      st.tw[n] = tw;
    }
  } else if (MODE == 1) {
    // S == 64: a strip's 16 queries share qy (q0 % 16 == 0); tile t is key row ky = t.
    //   th[c][t] = rel_h[qy - t + 63] . q_c ;  tw needs rel_w[qx_c - kx + 63], qx_c = q0x + c: build
    //   G[c][jj] = rel_w[q0x + jj] . q_c (jj < 80) in scratch (aliases the ring, not yet in use) and gather.
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      float *th = tabs + (size_t)(wave * NS + n) * 16 * S6D_GLB_THLD;
      float *G = reinterpret_cast<float *>(smem) + (size_t)wave * 16 * 80;      // per-wave scratch, reused by both strips
      const int q0y = div_S(p, q0[n]), q0x = q0[n] - q0y * p.S;
      build_table<HD, 4>(p.rel_h, q0y + 63, -1, st.qf[n], th, S6D_GLB_THLD, lane);
      build_table<HD, 5>(p.rel_w, q0x, 1, st.qf[n], G, 80, lane);
      // hipemu: wave rendezvous (same-wave LDS order; on the GPU the wave's LDS operations execute in program order)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) st.twr[n][sub * 4 + r] = G[c * 80 + (c + 63 - (sub * 16 + g * 4 + r))];
      // hipemu: wave rendezvous (the second strip's table overwrites G)
      thm[n] = th;
    }
  }
  const int ntile = p.T / 64;                       // launcher guarantees T % 64 == 0 for this kernel
  constexpr bool KSWZ = S6D_GLB_KSWZ != 0;
  StagerLinear<HD, WAVES * 64, KSWZ> sg;
  sg.init(p, b, head, tid);
  const size_t tstride = (size_t)64 * (size_t)p.tok_stride;
  sg.load(tstride, 0);
  __syncthreads();                                  // table scratch (aliasing the ring) fully consumed
  sg.store(Kbuf(0));
  __syncthreads();
  auto tile = [&](int t) {
    float thv[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) thv[n] = (MODE == 1) ? thm[n][c * S6D_GLB_THLD + t] : 0.f;
    if (!(kAbl & 2)) process_tile<HD, MODE, NS, KSWZ, S6D_GLB_PRIO != 0>(p, Kbuf(t & 1), Vbuf(t & 1), t * 64, st, thv, lane);
  };
  for (int t = 0; t + 1 < ntile; ++t) {                           // steady state: branch-free body
    if (!(kAbl & 1)) sg.load(tstride, t + 1);                     // flies under this tile's math
    tile(t);
    if (!(kAbl & 32)) sg.store(Kbuf((t & 1) ^ 1));                // ring slot last read in iteration t-1
    __syncthreads();
  }
  tile(ntile - 1);
#pragma unroll
  for (int n = 0; n < NS; ++n) store_strip<HD>(p, b, 0, 0, head, q0[n], st.lacc[n][0], st.oacc[n], lane);
}

// ---- global over the 64 x 64 grid, LDS-DMA staged: 8 waves x 32 queries per workgroup, K / V tiles through a 3-slot ring -------
// What the kernel above pays for besides its arithmetic (16 frames x 16 heads, same process, profiles/r02_attn_variants.txt):
// 2.22 ms as is, 1.75 ms without the staging of K / V tiles, 1.30 ms for the staging ALONE -- every 128-query workgroup pulls all
// 1.3 MB of its head's K and V through registers into LDS, one tile of look-ahead, 6 loads + 6 ds_write_b128 per thread and tile.
// Here: 256 queries per workgroup (half the K / V passes), tiles DMA'd straight into LDS (global_load_lds, 3 instructions per
// wave and tile, no staging registers, no ds_write) two tiles ahead of the arithmetic behind counted vmcnt waits and ONE raw
// s_barrier per tile, K rows chunk-swizzled on the source address (S6D_GLB_KSWZ above: conflict-free fragment reads), th tables
// at a 65-float row stride.  A ring slot is [K image 64 rows x (HDP + 8) | V image 64 x VROW]; every wave issues the same number
// of DMA instructions per tile (3 with 8 waves), so one vmcnt(3) means "my pieces of this tile landed".
// Measured in one process (16 frames, min of 3 x 20 launches): 1.73 ms (795 TFLOP/s) against 1.89 ms for the round-1 kernel,
// 1.81 ms for it with the two layout switches, 1.78 ms for it with 8 waves; 4 waves + 2 slots here: 1.77 ms.
// Where the rest goes (phase clocks of S6D_G64_TIMING, per tile and wave, older / younger half of the workgroup): barrier wait
// 850 / 180, DMA issue + th read 190 / 370, QK^T + scale + max 1260 / 1370, exp + pack 515 / 915, P V 545 / 530 -- about 3400
// cycles per tile for 96 MFMAs (1536 matrix-pipe cycles) and 2 x 158 VALU instructions on each SIMD.  The issue-rate probe
// (tools/probes/valu_rate.hip, profiles/r02_valu_rate.txt) says why: with two or more waves on a SIMD, MFMA and VALU issue time ADD
// (8 MFMA 57 ns, 48 v_fma 55 ns, 8 x (MFMA, 6 v_fma) 109 ns per wave; v_exp_f32 = 3 v_fma, packed fp32 ops = 1.8), so the
// softmax's 3.3 VALU instructions per MFMA cost about as much as the MFMAs themselves.  Tried on this kernel and dropped (no
// gain, same process): the tile as ONE interleaved stream -- K fragments read two 16-key blocks ahead, block i's scale / max
// beside block i + 1's MFMAs, P of keys 0..31 exponentiated speculatively against the old maximum beside the QK^T MFMAs, P of
// keys 32..63 beside the first P V MFMAs (1.74 ms, bit-identical output); static s_setprio 1 for the younger half (1.75 ms);
// rings of 2 and (without the th tables, as a timing probe) 6 slots (1.69 - 1.75 ms: the look-ahead is not what is missing).
#ifdef HIPEMU
#define S6D_ATTN_VMCNT(n) hipemu::vmcnt_wait(n)
#define S6D_ATTN_LGKM0()
#else
#define S6D_ATTN_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define S6D_ATTN_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
constexpr int G64_THLD = 65;
#ifndef S6D_G64_SLOTS
#define S6D_G64_SLOTS 3             // ring depth: tiles are DMA'd S6D_G64_SLOTS - 1 ahead of the arithmetic
#endif
#ifndef S6D_G64_WAVES
#define S6D_G64_WAVES 8
#endif
#ifndef S6D_G64_NOMAX
#define S6D_G64_NOMAX 1             // tiles after the first without a running maximum (process_tile_nomax); 0: round-4 arithmetic
#endif
#ifndef S6D_G64_NOMAX_PRIO
#define S6D_G64_NOMAX_PRIO 0        // s_setprio 1 around a no-max tile's matrix instructions
#endif
#ifndef S6D_G64_STATIC_PRIO
#define S6D_G64_STATIC_PRIO 0       // 1: the second-dispatched half of an 8-wave workgroup runs the whole tile loop at s_setprio 1
#endif

template <int HD, int WAVES_, int SLOTS_>
struct G64 {
  using C = Cfg<HD>;
  static constexpr int WAVES = WAVES_, NS = 2, SLOTS = SLOTS_;
  static constexpr int KCH = C::KROW / 8, VCH = C::VROW / 8;          // 16-byte chunks per K / V image row = KiB per image
  static constexpr int NPIECE = KCH + VCH;                            // 1-KiB DMA pieces per tile
  static constexpr int PW = (NPIECE + WAVES - 1) / WAVES;             // DMA instructions per wave and tile (a surplus one repeats the wave's previous piece)
  static constexpr int SLOT = NPIECE * 1024;
  static constexpr int RING = SLOTS * SLOT;
  static constexpr int TABS = WAVES * NS * 16 * G64_THLD * 4;
  static constexpr int LDS = RING + TABS + 16;                         // + the workgroup's "run the safe loop" flag
  static_assert(NPIECE > WAVES * (PW - 1) && PW >= 2, "every wave has a real piece to repeat");
  static_assert(RING >= WAVES * 16 * 80 * 4, "the prologue's per-wave scratch aliases the ring");
  static_assert(PW * (SLOTS - 2) <= 15 || SLOTS == 2, "vmcnt immediates used below");
};

// at most n tiles' DMA (PW instructions per wave and tile) may still be in flight
template <int PW>
__device__ __forceinline__ void g64_wait_tiles(int n) {
  switch (PW * n) {
#define S6D_G64_CASE(k) case k: S6D_ATTN_VMCNT(k); break;
    S6D_G64_CASE(0) S6D_G64_CASE(3) S6D_G64_CASE(5) S6D_G64_CASE(6) S6D_G64_CASE(9) S6D_G64_CASE(10) S6D_G64_CASE(12)
#undef S6D_G64_CASE
    default: S6D_ATTN_VMCNT(15); break;              // 15 or more (PW (SLOTS - 2) <= 15 is asserted)
  }
}

template <int HD, int WAVES, int SLOTS>
__global__ __launch_bounds__(WAVES * 64) void attn_global64_kernel(AttnParams p) {
  using C = Cfg<HD>;
  using G = G64<HD, WAVES, SLOTS>;
  constexpr int NS = G::NS, PW = G::PW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tabs = reinterpret_cast<float *>(smem + G::RING);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  // workgroup -> (image, head, query tile): the query tiles of one (image, head) stay on ONE XCD (K / V re-reads hit its L2)
  const int nqt = p.T / (WAVES * 16 * NS);
  const int nbh = p.B * p.nh;
  int id = blockIdx.x, qt, bh;
  if ((nbh & 7) == 0) {
    const int xcd = id & 7, loc = id >> 3;
    qt = loc % nqt;
    bh = (loc / nqt) * 8 + xcd;
  } else {
    qt = id % nqt;
    bh = id / nqt;
  }
  const int head = bh % p.nh, b = bh / p.nh;
  StripState<HD, NS> st;
  int q0[NS];
  float *thm[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    q0[n] = ((qt * WAVES + wave) * NS + n) * 16;
    load_q<HD>(p, b, 0, 0, head, q0[n], st.qf[n], lane);
    st.qy[n] = q0[n] >> 6;
    st.qx[n] = (q0[n] & 63) + c;
    st.m_run[n] = -1e30f;
    st.lacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) st.oacc[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.th[n] = nullptr;
    st.tw[n] = nullptr;
    // a strip's 16 queries share qy; tile t is key row ky = t:  th[c][t] = rel_h[qy - t + 63] . q_c, and
    // twr = rel_w[qx_c - kx + 63] . q_c gathered from G[c][jj] = rel_w[q0x + jj] . q_c (jj < 80; per-wave scratch in the ring)
    float *th = tabs + (size_t)(wave * NS + n) * 16 * G64_THLD;
    float *Gs = reinterpret_cast<float *>(smem) + (size_t)wave * 16 * 80;
    const int q0y = q0[n] >> 6, q0x = q0[n] & 63;
    build_table<HD, 4>(p.rel_h, q0y + 63, -1, st.qf[n], th, G64_THLD, lane);
    build_table<HD, 5>(p.rel_w, q0x, 1, st.qf[n], Gs, 80, lane);
    // hipemu: wave rendezvous (same-wave LDS order; on the GPU the wave's LDS operations execute in program order)
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) st.twr[n][sub * 4 + r] = Gs[c * 80 + (c + 63 - (sub * 16 + g * 4 + r))];
    // hipemu: wave rendezvous (the second strip's table overwrites the scratch)
    thm[n] = th;
  }
  // ---- this lane's DMA sources: piece q = wave + WAVES i of a tile is 64 consecutive chunks of the K image (q < KCH) or of the
  // V image; chunk -> (key row, part); zero-padded parts read a 16-byte zero; a wave without an i-th piece repeats its previous one
  // (same bytes to the same place), so that every wave has PW instructions per tile in flight and the waits count whole tiles
  const u16 *src[PW];
  long inc[PW];
  int dsto[PW];
  const long tstride = 64L * p.tok_stride;
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int q = wave + WAVES * i < G::NPIECE ? wave + WAVES * i : wave + WAVES * (i - 1);
    const bool isk = q < G::KCH;
    const int piece = isk ? q : q - G::KCH, rowch = isk ? G::KCH : G::VCH;
    const int j = piece * 64 + lane;
    const int row = j / rowch, pp = j - row * rowch;
    const int part = isk ? (pp < C::KPARTS ? pp ^ kswz(row) : C::KPARTS) : pp;
    const bool data = part * 8 < HD;
    src[i] = data ? qkv_at(p, (size_t)b * p.T + row, isk ? 1 : 2, head) + part * 8 : reinterpret_cast<const u16 *>(&g_win16_zero);
    inc[i] = data ? tstride : 0;
    dsto[i] = q * 1024;
  }
  auto issue = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      S6D_LDS(char) *dst = (S6D_LDS(char) *)smem + slot * G::SLOT + dsto[i];
      S6D_ATTN_DMA16(src[i], dst);
      src[i] += inc[i];
    }
  };
  const int ntile = p.T / 64;
  constexpr int D = SLOTS - 1;                      // look-ahead in tiles
  static_assert(D >= 1 && D <= 6, "ring depth");
  // column bias as the score chain's C operand (process_tile_nomax): tw / scale_log2, four key columns per register quad
  f32x4 cbias[NS][4];
  {
    const float inv = 1.0f / p.scale_log2;
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) cbias[n][sub][r] = st.twr[n][sub * 4 + r] * inv;
  }
  int *redo = reinterpret_cast<int *>(smem + G::RING + G::TABS);
  const u16 *src0[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) src0[i] = src[i];
  long long tk[6] = {0, 0, 0, 0, 0, 0}, tsum[5] = {0, 0, 0, 0, 0};
  int slot = 0;
  // everything of a tile in front of its arithmetic: this wave's pieces have landed, everybody's have (barrier; every wave is past
  // tile t - 1, so its slot is free), the DMA of tile t + D is issued into that slot, the row-bias words of the tile are read
  auto tile_head = [&](int t, float (&thv)[NS]) __attribute__((always_inline)) -> const u16 * {
    S6D_TICK(tk, 0);
    g64_wait_tiles<PW>(min(D - 1, ntile - 1 - t));
    S6D_ATTN_LGKM0();
    __builtin_amdgcn_s_barrier();
    S6D_TICK(tk, 1);
    if (t + D < ntile && !(kAbl & 1)) issue(slot >= 1 ? slot - 1 : SLOTS - 1);
#pragma unroll
    for (int n = 0; n < NS; ++n) thv[n] = thm[n][c * G64_THLD + t];
    S6D_TICK(tk, 2);
    const u16 *Kl = reinterpret_cast<const u16 *>(smem + slot * G::SLOT);
    slot = slot == SLOTS - 1 ? 0 : slot + 1;
    return Kl;
  };
  auto prime = [&]() __attribute__((always_inline)) {
    __syncthreads();                                // every wave is done with its scratch (it aliases the ring) / with the first pass
    if (tid == 0) *redo = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);
    slot = 0;
  };
  if (S6D_G64_STATIC_PRIO && WAVES == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);     // wave is wave-uniform (readfirstlane above)
  bool done = false;
  if (S6D_G64_NOMAX && !S6D_G64_TIMING) {
    // ---- pass A: tile 0 with the running-maximum arithmetic (it sets m_run), every later tile without a maximum ------------------
    prime();
    {
      float thv[NS];
      const u16 *Kl = tile_head(0, thv);
      process_tile<HD, 1, NS, true, S6D_GLB_PRIO != 0>(p, Kl, Kl + 64 * C::KROW, 0, st, thv, lane, tk);
    }
    for (int t = 1; t < ntile; ++t) {
      float thv[NS], nb[NS];
      const u16 *Kl = tile_head(t, thv);
#pragma unroll
      for (int n = 0; n < NS; ++n) nb[n] = thv[n] - st.m_run[n];
      process_tile_nomax<HD, NS, true, S6D_G64_NOMAX_PRIO != 0>(p, Kl, Kl + 64 * C::KROW, st, cbias, nb, lane);
    }
    // a row sum that left the comfortable range (or is inf / NaN): the whole workgroup repeats its tiles with the running maximum
    bool bad = false;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      bad |= !(st.lacc[n][0] < kNoMaxSumLimit);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= !(fabsf(st.oacc[n][dt][r]) < kNoMaxFinite);
    }
    if (__any(bad) && lane == 0) *redo = 1;
    __syncthreads();
    done = *redo == 0;
    if (!done) {
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        st.m_run[n] = -1e30f;
        st.lacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) st.oacc[n][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < PW; ++i) src[i] = src0[i];
    }
  }
  if (!done) {
    // ---- pass B: the round-4 kernel (every tile with the running maximum); the fallback of pass A ---------------------------------
    prime();
    for (int t = 0; t < ntile; ++t) {
      float thv[NS];
      const u16 *Kl = tile_head(t, thv);
      if (!(kAbl & 2)) process_tile<HD, 1, NS, true, S6D_GLB_PRIO != 0>(p, Kl, Kl + 64 * C::KROW, t * 64, st, thv, lane, tk);
      S6D_TICK(tk, 5);
      if (S6D_G64_TIMING)
#pragma unroll
        for (int i = 0; i < 5; ++i) tsum[i] += tk[i + 1] - tk[i];
    }
  }
#pragma unroll
  for (int n = 0; n < NS; ++n) store_strip<HD>(p, b, 0, 0, head, q0[n], st.lacc[n][0], st.oacc[n], lane);
  if (S6D_G64_TIMING && blockIdx.x == 8 && lane == 0) {             // the caller of a probe build leaves 4 KiB behind the output
    long long *dbg = reinterpret_cast<long long *>(p.out + (size_t)p.B * p.T * p.nh * HD);
#pragma unroll
    for (int i = 0; i < 5; ++i) dbg[wave * 5 + i] = tsum[i];
  }
}

template <int HD>
static int launch_attn(AttnParams p, hipStream_t st) {
  using C = Cfg<HD>;
  const bool bias = p.rel_h != nullptr;
  if (p.ws > 0 && bias && p.S <= 16) {
    constexpr int WAVES = 8;                                      // rows (wave, wave + 8): two query rows per wave share every
                                                                  // K / V fragment read (measured: 8x2 rows 0.23 ms vs 14x1 rows 0.25 ms per 8 frames)
    const int SR = (p.S + 1) & ~1;
    // persistent, LDS-DMA double-buffered kernel: two query rows per wave (S > 8), both images + the rel tables inside 160 KiB;
    // windows of up to 8 rows take the one-item kernel below
    if (p.S > WAVES) {
      const int kin = (SR * 16 * (HD + 8) / 8 + 63) / 64, vin = (SR * 16 * C::VROW / 8 + 63) / 64;
      const size_t lds = (size_t)2 * (kin + vin) * 1024 + (size_t)2 * 32 * C::HDP * 2;
      const int nitems = p.B * p.nwy * p.nwx * p.nh;
      if (lds <= 160 * 1024 && (size_t)WAVES * 16 * W16_LT * 4 <= (size_t)vin * 1024) {
        int grid = nitems < 256 ? nitems : 256;                   // one persistent workgroup per CU
        if (s6d::g_s6d_persistent_grid_limit > 0 && s6d::g_s6d_persistent_grid_limit < grid) grid = s6d::g_s6d_persistent_grid_limit;   // set through the ABI (tests)
        if (grid >= 8) grid &= ~7;                                // whole XCD rounds: workgroup j keeps to XCD j % 8, like its items
        if (p.S == 14) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_window16p_kernel<HD, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL((attn_window16p_kernel<HD, true>), dim3(grid), dim3(512), lds, st, p, nitems);
        } else {
          (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_window16p_kernel<HD, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL((attn_window16p_kernel<HD, false>), dim3(grid), dim3(512), lds, st, p, nitems);
        }
        return launch_status();
      }
    }
    const size_t lds = (size_t)SR * 16 * (C::KROW + C::VROW) * 2 + (size_t)WAVES * 16 * W16_LT * 4;
    if (lds > 160 * 1024) return S6D_EUNSUPPORTED;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_window16_kernel<HD, WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned grid = (unsigned)(p.B * p.nwy * p.nwx * p.nh);
    hipLaunchKernelGGL((attn_window16_kernel<HD, WAVES>), dim3(grid), dim3(WAVES * 64), lds, st, p);
  } else if (p.ws > 0) {
    constexpr int WAVES = 8;
    const int ntile = (p.T + 63) / 64;
    const size_t lds = bias ? (size_t)ntile * 64 * (C::KROW + C::VROW) * 2 + (size_t)WAVES * 2 * 16 * p.LT * 4
                            : ((size_t)win_seq_krows(p.T) * C::KROW + (size_t)win_seq_vrows(p.T) * C::VROW) * 2;
    if (lds > 160 * 1024) return S6D_EUNSUPPORTED;
    const unsigned grid = (unsigned)(p.B * p.nwy * p.nwx * p.nh);
    if (bias) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_window_kernel<HD, WAVES, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attn_window_kernel<HD, WAVES, true>), dim3(grid), dim3(WAVES * 64), lds, st, p);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_window_kernel<HD, WAVES, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attn_window_kernel<HD, WAVES, false>), dim3(grid), dim3(WAVES * 64), lds, st, p);
    }
  } else {
    if (p.T % 64 != 0) return S6D_EUNSUPPORTED;                  // global grids: 16x16, 32x32, 64x64 ...
    constexpr int WAVES = S6D_GLB_WAVES;
    constexpr int NS = 2;
    const size_t ring = (size_t)2 * 64 * (C::KROW + C::VROW) * 2;
    const int nqt = (p.T + WAVES * 16 * NS - 1) / (WAVES * 16 * NS);
    const unsigned grid = (unsigned)(p.B * p.nh * nqt);
#define S6D_GLB(MODE, LDS)                                                                                     \
  do {                                                                                                         \
    if ((LDS) > 160 * 1024) return S6D_EUNSUPPORTED;                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_global_kernel<HD, WAVES, MODE>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));                         \
    hipLaunchKernelGGL((attn_global_kernel<HD, WAVES, MODE>), dim3(grid), dim3(WAVES * 64), (LDS), st, p);     \
  } while (0)
    if (bias && p.S == 64) {
      using G = G64<HD, S6D_G64_WAVES, S6D_G64_SLOTS>;
      static_assert(G::LDS <= 160 * 1024, "ring + tables fit the CU's LDS");
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_global64_kernel<HD, G::WAVES, G::SLOTS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
      hipLaunchKernelGGL((attn_global64_kernel<HD, G::WAVES, G::SLOTS>), dim3((unsigned)(p.B * p.nh * (p.T / (G::WAVES * 32)))),
                         dim3(G::WAVES * 64), G::LDS, st, p);
      return launch_status();
    }
    if (!bias) {
      S6D_GLB(2, ring);
    } else {
      S6D_GLB(0, ring + (size_t)WAVES * NS * 2 * 16 * p.LT * 4);
    }
#undef S6D_GLB
  }
  return launch_status();
}

}  // namespace S6D_ATTN_NS

using namespace S6D_ATTN_NS;

#if !S6D_ATTN_F16
extern "C" long s6d_win_attention_scratch_bytes(int H, int window, int head_dim) {
  const int S = window ? window : H;
  const int LT = ((2 * S - 1) + 15) / 16 * 16, HDP = (head_dim + 31) / 32 * 32;
  return 2L * (LT + 16) * HDP * 2;
}

extern "C" int s6d_win_attention_layout_bf16(const void *qkv, int head_major, const void *qkv_bias, const void *rel_h, const void *rel_w,
                                             int B, int H, int W, int num_heads, int head_dim, int window, float scale,
                                             void *rel_scratch, void *out, void *stream);

extern "C" int s6d_win_attention_bf16(const void *qkv, const void *qkv_bias, const void *rel_h, const void *rel_w,
                                      int B, int H, int W, int num_heads, int head_dim, int window, float scale,
                                      void *rel_scratch, void *out, void *stream) {
  return s6d_win_attention_layout_bf16(qkv, 0, qkv_bias, rel_h, rel_w, B, H, W, num_heads, head_dim, window, scale, rel_scratch, out,
                                       stream);
}

// prepadded != 0: rel_scratch already holds the padded tables (s6d_win_attention_pad_rel_bf16, made once per weight version: the
// padding is a function of the two weight tables only and ran as a 5-us launch in front of every one of the 64 attention launches
// of a step); rel_h / rel_w are then only tested for presence.
static int win_attention_impl(const void *qkv, int head_major, const void *qkv_bias, const void *rel_h, const void *rel_w,
                              int B, int H, int W, int num_heads, int head_dim, int window, float scale,
                              void *rel_scratch, int prepadded, void *out, void *stream) {
  if (B < 0 || H <= 0 || W <= 0 || num_heads <= 0 || window < 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!qkv || !qkv_bias || !out || ((rel_h == nullptr) != (rel_w == nullptr))) return S6D_EINVAL;
  if (window == 0 && H != W) return S6D_EUNSUPPORTED;
  if (window == 0 && (H * W) % 64 != 0) window = H;   // small odd grids: one all-resident "window" = whole grid
  AttnParams p;
  p.qkv = (const u16 *)qkv; p.qkv_bias = (const u16 *)qkv_bias;
  p.rel_h = (const u16 *)rel_h; p.rel_w = (const u16 *)rel_w; p.out = (u16 *)out;
  p.B = B; p.H = H; p.W = W; p.nh = num_heads; p.ws = window;
  p.S = window ? window : H; p.T = p.S * p.S;
  p.nwx = window ? (W + window - 1) / window : 1;
  p.nwy = window ? (H + window - 1) / window : 1;
  p.LT = ((2 * p.S - 1) + 15) / 16 * 16;
  p.magicS = (unsigned)(((1ull << 32) + (unsigned)p.S - 1) / (unsigned)p.S);
  p.scale_log2 = scale * kLog2e;
  if (head_major) {                              // (3, nh, B H W, hd): the qkv GEMM's column-block output
    const long plane = (long)B * H * W * head_dim;
    p.tok_stride = head_dim; p.head_stride = plane; p.which_stride = (long)num_heads * plane;
  } else {                                       // (B, H, W, 3, nh, hd): the raw Linear output
    p.tok_stride = 3L * num_heads * head_dim; p.which_stride = (long)num_heads * head_dim; p.head_stride = head_dim;
  }
  hipStream_t st = as_stream(stream);
  if (rel_h) {                                   // zero-padded table copies: unconditional loads in the kernels
    if (!rel_scratch) return S6D_EINVAL;
    const int HDP = (head_dim + 31) / 32 * 32, rows = p.LT + 16;
    if (!prepadded)
      hipLaunchKernelGGL(pad_rel_kernel, dim3(16), dim3(256), 0, st, p.rel_h, p.rel_w, 2 * p.S - 1, head_dim, HDP, rows,
                         (u16 *)rel_scratch);
    p.rel_h = (const u16 *)rel_scratch;
    p.rel_w = p.rel_h + (size_t)rows * HDP;
  }
  switch (head_dim) {
    case 80: return launch_attn<80>(p, st);
    case 64: return launch_attn<64>(p, st);
    default: return S6D_EUNSUPPORTED;
  }
}

extern "C" int s6d_win_attention_layout_bf16(const void *qkv, int head_major, const void *qkv_bias, const void *rel_h, const void *rel_w,
                                             int B, int H, int W, int num_heads, int head_dim, int window, float scale,
                                             void *rel_scratch, void *out, void *stream) {
  return win_attention_impl(qkv, head_major, qkv_bias, rel_h, rel_w, B, H, W, num_heads, head_dim, window, scale, rel_scratch, 0, out, stream);
}

extern "C" int s6d_win_attention_pad_rel_bf16(const void *rel_h, const void *rel_w, int H, int window, int head_dim, void *rel_padded,
                                              void *stream) {
  if (!rel_h || !rel_w || !rel_padded || H <= 0 || window < 0 || head_dim <= 0) return S6D_EINVAL;
  const int S = window ? window : H, LT = ((2 * S - 1) + 15) / 16 * 16;
  const int HDP = (head_dim + 31) / 32 * 32, rows = LT + 16;
  hipLaunchKernelGGL(pad_rel_kernel, dim3(16), dim3(256), 0, as_stream(stream), (const u16 *)rel_h, (const u16 *)rel_w, 2 * S - 1, head_dim,
                     HDP, rows, (u16 *)rel_padded);
  return launch_status();
}

extern "C" int s6d_win_attention_prepadded_bf16(const void *qkv, int head_major, const void *qkv_bias, const void *rel_padded, int B, int H,
                                                int W, int num_heads, int head_dim, int window, float scale, void *out, void *stream) {
  if (!rel_padded) return S6D_EINVAL;
  return win_attention_impl(qkv, head_major, qkv_bias, rel_padded, rel_padded, B, H, W, num_heads, head_dim, window, scale,
                            const_cast<void *>(rel_padded), 1, out, stream);
}

#endif  // !S6D_ATTN_F16

#if S6D_ATTN_F16
#define S6D_SEQ_ATTENTION s6d_seq_attention_f16
#else
#define S6D_SEQ_ATTENTION s6d_seq_attention_bf16
#endif
#if S6D_ATTN_F16
#define S6D_SEQ_ATTENTION_STRIDED s6d_seq_attention_strided_f16
#else
#define S6D_SEQ_ATTENTION_STRIDED s6d_seq_attention_strided_bf16
#endif
// q / k / v element (sequence b, token n, which, head h, d) sits at qkv + (b N + n) tok_stride + which which_stride + h head_stride + d:
//   token-major (the raw Linear output (B, N, 3, nh, hd)):  tok_stride = 3 nh hd, which_stride = nh hd, head_stride = hd
//   head-major  ((3, nh, B N, hd), the qkv GEMM's column-block epilogue):  tok_stride = hd, head_stride = B N hd, which_stride = nh B N hd
// Head-major makes the K / V rows of one (sequence, head) ONE contiguous run (257 x 128 B = 32 KB) instead of 257 pieces of 128 B
// strided by 6 KB: measured on the DINOv2 shape, the fetch of the token-major pieces ALONE costs 136 us per launch (2.3 TB/s).
extern "C" int S6D_SEQ_ATTENTION_STRIDED(const void *qkv, long tok_stride, long which_stride, long head_stride, int B, int N,
                                         int num_heads, int head_dim, float scale, void *out, void *stream) {
  if (B < 0 || N <= 0 || num_heads <= 0 || head_dim <= 0) return S6D_EINVAL;
  if (tok_stride < head_dim || (tok_stride % 8) || (which_stride % 8) || (head_stride % 8)) return S6D_EINVAL;   // 16-byte chunks
  if (B == 0) return S6D_OK;
  if (!qkv || !out || ((uintptr_t)qkv & 15)) return S6D_EINVAL;
  // a 1 x N "image" attended as ONE all-resident window of N key slots, no positional bias
  AttnParams p;
  p.qkv = (const u16 *)qkv; p.qkv_bias = (const u16 *)qkv;        // never read: every slot < N is in-image
  p.rel_h = nullptr; p.rel_w = nullptr; p.out = (u16 *)out;
  p.B = B; p.H = 1; p.W = N; p.nh = num_heads; p.ws = N;
  p.S = N; p.T = N; p.nwx = 1; p.nwy = 1; p.LT = 16;
  p.magicS = (unsigned)(((1ull << 32) + (unsigned)N - 1) / (unsigned)N);
  p.scale_log2 = scale * kLog2e;
  p.tok_stride = tok_stride; p.which_stride = which_stride; p.head_stride = head_stride;
  hipStream_t st = as_stream(stream);
  switch (head_dim) {
    case 80: return launch_attn<80>(p, st);
    case 64: return launch_attn<64>(p, st);
    default: return S6D_EUNSUPPORTED;
  }
}

extern "C" int S6D_SEQ_ATTENTION(const void *qkv, int B, int N, int num_heads, int head_dim, float scale, void *out,
                                 void *stream) {
  if (num_heads <= 0 || head_dim <= 0) return S6D_EINVAL;
  return S6D_SEQ_ATTENTION_STRIDED(qkv, 3L * num_heads * head_dim, (long)num_heads * head_dim, head_dim, B, N, num_heads, head_dim,
                                   scale, out, stream);
}
