// bf16 GEMM with fused epilogue for the Linear layers of the ViTs on the hot path (gfx950):  C = epi(A W^T + bias).
//
// Reference (the nn.Linear calls this replaces; the reference runs them through cuBLAS and a separate GELU kernel):
//   segment_anything/modeling/common.py:13-28         MLPBlock  lin1 (1280 -> 5120) + exact GELU, lin2 (5120 -> 1280)
//   segment_anything/modeling/image_encoder.py:224-240 Attention qkv (1280 -> 3840), proj (1280 -> 1280)
//   segment_anything/modeling/image_encoder.py:90-104  neck 1x1 conv (= Linear 1280 -> 256 on channels-last tokens)
//   Pose_Estimation_Model/model/feature_extraction.py:17-35 (timm ViT-B blocks), Instance_Segmentation_Model/model/layers/
//   {attention,mlp}.py (DINOv2 ViT-L blocks) -- same Linear / GELU statements.
// A (M,K) activations and W (N,K) nn.Linear weight are both K-contiguous, so both operands are read the same way.
//
// Structure (one 512-thread workgroup per CU, 256 x 256 output tile, K step 64, persistent over its share of the tiles):
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) in HALF-TILES of 128 rows x 64 k (16 KiB).  All of
//     the CU's 160 KiB LDS is one ring of 10 half-tile slots; half-tiles are issued in stream order B0 B1 A0 A1 per K tile,
//     one per phase, 6 phases (1.5 K tiles) ahead of the phase that first reads them, and the stream runs across output-tile
//     boundaries, so the next tile's operands arrive while this tile's epilogue runs.  Waits are COUNTED (vmcnt(8) / (6)):
//     the queue never drains inside the loop.
//   * LDS image of a half-tile: [128 rows][8 chunks of 16 B], chunk position c' = c ^ ((row >> 1) & 7).  The DMA writes
//     lane-linear, so the permutation is applied to each lane's SOURCE address (it stays inside one 128-byte line) and again
//     on the fragment read: every ds_read_b128 lane group then covers 16 distinct 16-byte bank slots (conflict-free).
//   * 8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 of the tile as 4 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16).  The
//     product is formed TRANSPOSED (W fragment as the A operand, activation fragment as B): a lane then holds one output
//     row and 4 consecutive output columns per register quad, which makes the epilogue stores 16 B per lane.
//   * a K tile is 4 phases (one 64 x 32 quadrant of the wave's tile over the whole K step each); the two wave groups
//     (M halves) run one barrier apart, so on every SIMD one wave is in its matrix segment (8 MFMAs) while the other issues
//     its fragment reads and DMA: phase = { ds_read, DMA issue, [counted wait], barrier, 8 x MFMA, barrier }.
//   * EPI 2 / 3 / 4 (round 3): the residual add and the LayerNorm of a ViT block folded into the GEMMs on either side of them
//     (x + proj(..), x + lin2(..); LN1 -> qkv, LN2 -> lin1 + GELU): see "Residual add and LayerNorm around the GEMM" below.
//   * epilogue in registers: bias is the accumulator's initial value, erf-form GELU as relu(x) - |x| 2^P(|x|) (gelu_erf below:
//     relative error 6e-6), round to bf16,
//     v_permlane32_swap pairs the two lane halves into 16-byte row segments.
#include "s6d_common.h"
#include "s6d_gemm_params.h"
#include <stdlib.h>

namespace s6d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define S6D_LDS(T) __attribute__((address_space(3))) T
#define S6D_GLOBAL(T) __attribute__((address_space(1))) T

#ifdef HIPEMU
#define S6D_VMCNT(n) hipemu::vmcnt_wait(n)
#define S6D_CONST(T) T
#else
#define S6D_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define S6D_CONST(T) __attribute__((address_space(4))) T   // uniform address + constant address space = s_load (lgkmcnt, not vmcnt)
#endif
// An MFMA is a pure register operation: neither s_barrier nor sched_barrier orders it at instruction-selection time (it has
// no chain), so the matrix segment is tied down by data: an empty asm makes the operands opaque after the opening barrier
// (no MFMA can be placed above it) and the accumulators opaque before the closing one (none can sink below it).
#ifdef HIPEMU
#define S6D_PIN(x)
#else
#define S6D_PIN(x) asm volatile("" : "+v"(x))
#endif
// hipcc treats s_barrier as a memory fence only: register-only MFMAs drift across it (and past s_setprio) unless pinned
#define S6D_BARRIER()                     \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// Profiling switches, COMPILE-time (tools/gemm_variants.sh builds one library per setting; the product library has none):
//   S6D_GEMM_ABLATE bit 1 = no LDS-DMA, 2 = no MFMA (fragment reads stay), 4 = no epilogue stores;  S6D_GEMM_NOPRIO = no s_setprio
#ifndef S6D_GEMM_ABLATE
#define S6D_GEMM_ABLATE 0
#endif
#ifdef S6D_GEMM_NOPRIO
#define S6D_SETPRIO(n)
#else
#define S6D_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

// Epilogue layout.  Measured on the ViT-H shapes (profiles/r02_gemm_variants.json): the same kernel without its stores runs
// qkv / proj 25 % faster -- the stores, not the matrix work, are the largest loss at K = 1280.  In the accumulator layout a lane
// owns ONE output row, so a 16-byte-per-lane store instruction touches 64 different rows: 64 write requests of 16 B.  With
// S6D_GEMM_QT (default) the four lanes of a quad exchange their four 16-byte chunks (a 4 x 4 transpose by DPP quad_perm) so
// that a quad holds 64 contiguous bytes of one row: 16 requests of 64 B per instruction.  Two things make that possible without
// a second exchange: the W rows are fed to the matrix instruction in a permuted order (lane half hb owns columns 32 hb ..
// 32 hb + 31 of the wave's 64, tile nt the 16-column group nt of each half), which also removes the permlane32 swaps.
// Tried and dropped (same file): the epilogue rolled quadrant by quadrant into the next tile's first K tile (-3 %: its VALU work
// lengthens the load segments) and start-time staggering of the workgroups (-4 %).
#ifndef S6D_GEMM_QT
#define S6D_GEMM_QT 1
#endif
// Main loop.  S6D_GEMM_PH2 = 1: TWO phases per K tile (16-MFMA segments: half the workgroup barriers per MFMA; a phase's load segment
// = 16 / 8 fragment reads + 4 LDS-DMA pieces against 512 cycles of the partner's matrix segment) instead of the four 8-MFMA phases of
// the template (S6D_GEMM_PH2 = 0).  Measured in one process, two interleaved rounds (profiles/r02_gemm_variants_ph2.json): +3..4 % on
// all four ViT-H shapes (qkv 1161 -> 1202, proj 1115 -> 1140, lin1 + GELU 1025 -> 1047, lin2 1247 -> 1285 TFLOP/s): the default.
#ifndef S6D_GEMM_PH2
#define S6D_GEMM_PH2 1
#endif
#ifndef S6D_GEMM_AUX_A
#define S6D_GEMM_AUX_A 0
#endif
#ifndef S6D_GEMM_AUX_B
#define S6D_GEMM_AUX_B 0
#endif
// Also tried and dropped: draining the ring (vmcnt(0)) before the stores so that no counted wait sits behind them for 7 phases --
// the single vector-memory counter makes a counted wait behind 16 stores wait for their acknowledgement -- measured -8 % (the drain
// itself exposes a load latency per tile and the store cost did not move: profiles/r02_gemm_variants_qt_drain.json).

// Residual add and LayerNorm around the GEMM (round 3).
//   EPI 2  C = A W^T + bias + R, summed in fp32 and rounded to bf16 once (the separate add rounded the product first).  The
//          accumulators START at bias + residual.  A tile's residual values (128 per lane, in the accumulator layout) are fetched
//          during the PREVIOUS tile's epilogue, strip by strip into the accumulator registers that strip has just stored, and are
//          first read after that epilogue, at the point where the plain kernel reads its bias (a drain it spends anyway).  Tried and
//          measured first: (a) residual loads consumed inside the same epilogue wait for the tile's own stores on the single in-order
//          memory counter (proj + 24 %, lin2 + 8..14 %, profiles/r03_gemm_residual_epilogue.txt); (b) the residual tile as 4 extra
//          K tiles against an identity weight operand costs those K tiles plus an HBM-latency-bound stream (proj + 22 %,
//          lin2 + 11 %; neither skipping the idle wave columns' matrix work nor moving the K tiles to the head of the tile
//          helped: profiles/r03_lnfold.txt).  Optionally the epilogue emits, per row and 32-column group, (sum, sum of squared
//          deviations from the group mean) of the fp32 results -- partial LayerNorm statistics, combined by
//          ln_stats_finalize_kernel (csrc/s6d_norm.hip).
//   EPI 3/4  C = [GELU] LN(A) W^T + bias with the LayerNorm folded:  LN(a) W^T = rstd_m (a W'^T - mu_m s_n) + b'_n  with
//          W' = gamma o W (bf16), s_n = sum_k W'_nk, b' = bias + W beta.  The kernel multiplies the RAW residual stream by W';
//          the accumulators start at sigma_m b'_n - mu_m s_n (per-row statistics and both column vectors are read at the tile
//          start, where the bias read of the plain form sits) and the epilogue multiplies by rstd_m = 1 / sigma_m:  the
//          normalised activations are never written (and never rounded to bf16).


constexpr int kSlot = 16384;  // one half-tile: 128 rows x 64 k bf16
constexpr int kRing = 10;     // slots: all 160 KiB of the CU

extern __shared__ __attribute__((aligned(16))) char gemm_smem[];

__device__ __forceinline__ int ring(int s) { return s >= kRing ? s - kRing : s; }
__device__ __forceinline__ int ring5(int s) { return s >= 5 ? s - 5 : s; }

// gelu_erf(): csrc/s6d_common.h (erf-form GELU as relu(x) - |x| 2^P(|x|), 6e-6 relative)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  union { __bf16 b; u16 u; } a, c;
  a.b = (__bf16)lo;                                                     // round-to-nearest-even
  c.b = (__bf16)hi;
  return (unsigned)a.u | ((unsigned)c.u << 16);
}

// DT = 0: bf16 operands, K step 64 (4 x v_mfma_f32_32x32x16_bf16 per 32 x 32 tile and K tile).
// DT = 1 (round 3, BASELINE configs[4]): OCP fp8 e4m3 operands, K step 128 -- the SAME 128-byte rows, ring, swizzle and DMA
// stream -- and 2 x v_mfma_scale_f32_32x32x64_f8f6f4 per tile and K tile: twice the product per byte staged and per matrix
// cycle.  The per-token and per-output-channel scales are powers of two (E8M0 bytes) and ride in the instruction's hardware
// block-scale operands: a lane's fragment is one row's 32 k values, so "block scale" = that row's scale, constant over K.  The
// accumulators therefore hold the true product and the epilogue (bias init, GELU, quad transpose) is the bf16 one, untouched.
// Both operands are fetched with the same (lane >> 5, byte) -> k map, so the product does not depend on the instruction's
// internal k order.
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// DT = 2 (round 3): IEEE half operands and output -- the bf16 kernel with v_mfma_f32_32x32x16_f16 and an f16 pack, everything else
// (2-byte elements: rows, ring, swizzle, epilogue layout) shared.  The PEM's ViT-B runs on it: half's 11-bit significand keeps
// the features within 1e-3 of the fp32 extractor's (bf16: 7.6e-3), which is what the 1e-3 mm translation bar needs.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned pack_f16(float lo, float hi) {
  union { _Float16 h; u16 u; } a, c;
  a.h = (_Float16)lo;                                                   // round-to-nearest-even
  c.h = (_Float16)hi;
  return (unsigned)a.u | ((unsigned)c.u << 16);
}
// bf16: ONE v_cvt_pk_bf16_f32 per dword (round 6; the scalar conversions compile to a conversion per value and an SDWA or).  IEEE half
// stays value by value: v_cvt_pk_f16_f32 differs from v_cvt_f16_f32 on results below the smallest normal half.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ unsigned pack_out(float lo, float hi) {
#ifdef HIPEMU
  return DT == 2 ? pack_f16(lo, hi) : pack_bf16(lo, hi);
#else
  if constexpr (DT == 2) return pack_f16(lo, hi);
  else return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
#endif
}

// AMX (DT = 1 only; round 4): the activations carry MX block scales -- one E8M0 byte per row and 32 k (GemmParams::sa_mx) instead of
// one per row.  The matrix instruction takes exactly that: a lane's scale byte applies to its own 32-k block and op_sel picks the
// byte of the scale VGPR (measured: tools/probes/mx_scale_probe.hip, mx_scale_probe2.hip: the byte of lanes 0-31 scales the FIRST 16
// bytes of both lane halves, the byte of lanes 32-63 the second 16 -- hence the chunk map of the fragment reads below).  A K tile is
// 128 bytes = four blocks = one dword of scales per row; lane half hb SUPPLIES the scales of blocks hb (first MFMA of the tile) and
// 2 + hb (second), so the dword is shifted right by 8 hb once and the two instructions use op_sel 0 and 2.  The dword of K tile g + 1 is requested at the start of K tile g, in FRONT of that
// phase's LDS-DMA issues: it is older than the eight DMA pieces the counted wait leaves in flight, so "K tile g + 1 has landed"
// covers it and the counts stay as they are.  Rows past M are not clamped: the scale array is readable for whole 256-row tiles.
// EPI 5 (DT = 1): C = e4m3(GELU(A W^T + bias)) with MX scales -- in the quad-transposed accumulator layout ONE lane holds a whole
// 32-column block of its row (columns 32 h + {0..31} of the wave's 64), so the block maximum, the scale and the 32 bytes are
// lane-local: no exchange, two 16-byte stores and one scale byte per lane and m tile.
template <int EPI, bool HAS_BIAS, int DT, bool AMX = false>
__device__ __forceinline__ void gemm_body(const GemmParams &p) {
  static_assert(!AMX || DT == 1, "MX block scales belong to the fp8 operands");
  static_assert(EPI != 5 || (DT == 1 && S6D_GEMM_QT), "the MX output is the fp8 kernel's, on the quad-transposed layout");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;       // M half / N quarter of the 256 x 256 tile

  // ---- tile schedule.  Workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed only): XCD x takes a
  // contiguous range of the logical tile order, its workgroups walk that range with stride (workgroups per XCD), and
  // the logical order goes down GM m-tiles before moving to the next n-tile, so the tiles one XCD works on at a time share
  // A panels and W panels through its L2.
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int tq = p.ntiles >> 3, tr = p.ntiles & 7;
  const int first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int cnt = tq + (xcd < tr ? 1 : 0);
  if (bi >= cnt) return;
  const int my_tiles = (cnt - bi + bpx - 1) / bpx;
  const int G = my_tiles * p.nk;                 // K tiles this workgroup streams through

  auto tile_mn = [&](int j, int &m0, int &n0) __attribute__((always_inline)) {
    const int L = first + bi + j * bpx;
    const int tpg = p.GM * p.NT;
    const int grp = L / tpg, rem = L - grp * tpg;
    const int mf = grp * p.GM;
    const int gs = min(p.GM, p.MT - mf);
    const int nn = rem / gs;
    m0 = (mf + rem - nn * gs) * 256;
    n0 = nn * 256;
  };

  // ---- staging geometry: wave w fills rows [16 w, 16 w + 16) of a half-tile in two 1-KiB pieces (8 rows x 128 B each);
  // lane -> row (lane >> 3), chunk position (lane & 7); the chunk it fetches is position ^ ((row >> 1) & 7)
  const int srow = wave * 16 + (lane >> 3);
  const unsigned sc0 = (unsigned)(((lane & 7) ^ (lane >> 4)) << 4);     // piece 0; piece 1 (rows + 8): sc0 ^ 64
  const unsigned sd1 = (sc0 ^ 64u) - sc0;
  unsigned a_off[2][2], b_off = 0;
  int ia_kt = 0, ia_tile = 0, ib_kt = 0, ib_tile = 0;                   // issue cursors (K tile within the output tile, tile)
  auto set_a = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_mn(tile, m0, n0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)                                       // rows past M re-read row M - 1 (never stored)
        a_off[a][i] = (unsigned)min(m0 + a * 128 + srow + 8 * i, p.M - 1) * p.lda2 + (sc0 ^ (unsigned)(i * 64));
  };
  auto set_b = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_mn(tile, m0, n0);
    b_off = (unsigned)(n0 + srow) * p.ldw2 + sc0;
  };
  // S6D_GEMM_AUX_A / _B: cache-policy bits of the LDS-DMA of the activation / weight stream (2 = nt, "streaming"): experiment knobs,
  // see profiles/r03_lnfold.txt for what they measured
  auto dma = [&](const u16 *base, unsigned off, int slot, int piece, bool is_b = false) __attribute__((always_inline)) {
    S6D_LDS(char) *dst = (S6D_LDS(char) *)gemm_smem + slot * kSlot + (wave * 2 + piece) * 1024;
    if (S6D_GEMM_ABLATE & 1) return;
    if (is_b)
      __builtin_amdgcn_global_load_lds((const S6D_GLOBAL(void) *)((const char *)base + off), dst, 16, 0, S6D_GEMM_AUX_B);
    else
      __builtin_amdgcn_global_load_lds((const S6D_GLOBAL(void) *)((const char *)base + off), dst, 16, 0, S6D_GEMM_AUX_A);
  };
  auto issue_b = [&](int half, int slot) __attribute__((always_inline)) {
    const unsigned o = b_off + (unsigned)ib_kt * 128u + (unsigned)half * 128u * p.ldw2;
    dma(p.W, o, slot, 0, true);
    dma(p.W, o + 8u * p.ldw2 + sd1, slot, 1, true);
    if (half == 1 && ++ib_kt == p.nk) {
      ib_kt = 0;
      if (++ib_tile < my_tiles) set_b(ib_tile);
    }
  };
  auto issue_a = [&](int half, int slot) __attribute__((always_inline)) {
    const unsigned k = (unsigned)ia_kt * 128u;
    dma(p.A, a_off[half][0] + k, slot, 0);
    dma(p.A, a_off[half][1] + k, slot, 1);
    if (half == 1 && ++ia_kt == p.nk) {
      ia_kt = 0;
      if (++ia_tile < my_tiles) set_a(ia_tile);
    }
  };

  // ---- fragment geometry (32x32x16: lane -> row lane & 31, 8 consecutive k at 8 (lane >> 5) of the 16-wide k step)
  unsigned foff[4];
  {
    const int sw = (lane >> 1) & 7, hb = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // 16-byte chunk of the row this lane reads at step ks.  bf16: k step ks, k half hb.  fp8: MFMA ks >> 1 covers 64 bytes = four
      // chunks; the instruction's two 32-element SCALE blocks are (first 16 bytes of lanes 0-31, first 16 bytes of lanes 32-63) and
      // (second 16 bytes of both), scaled by the byte of lanes 0-31 and of lanes 32-63 respectively (measured:
      // tools/probes/mx_scale_probe2.hip) -- so for a scale block to be 32 CONSECUTIVE k (the MX format) lane half hb takes chunks
      // hb and 2 + hb.  (Rounds 3's map, chunks 2 hb and 2 hb + 1, is equivalent only while both blocks of a row share one scale.)
      const int cid = DT == 1 ? (4 * (ks >> 1) + hb + 2 * (ks & 1)) : ((2 * ks) | hb);
      foff[ks] = (unsigned)((lane & 31) * 128 + ((cid ^ sw) << 4));
    }
  }
  const unsigned brow = (unsigned)((wc & 1) * 64 * 128);                // this wave's 64 W rows inside its B half-tile
  auto frag = [&](int slot, unsigned rowbytes, int ks) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8 *>(gemm_smem + slot * kSlot + rowbytes + foff[ks]);
  };
  // W fragment of n tile nt.  S6D_GEMM_QT: matrix row a (= A-operand lane, a = 4 h + 8 qd + e with h = (a >> 2) & 1) is fed W row
  // 32 h + 16 nt + 4 qd + e of the wave's 64, so that in the accumulators lane half h owns columns 32 h + 16 nt + {0..15}.
  // (rows mod 16 stay distinct inside every ds_read_b128 lane group: the swizzle stays conflict-free)
  unsigned wfo[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int a = lane & 31;
    const int row = S6D_GEMM_QT ? 32 * ((a >> 2) & 1) + 16 * nt + 4 * (a >> 3) + (a & 3) : 32 * nt + a;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int hb = lane >> 5;
      const int cid = DT == 1 ? (4 * (ks >> 1) + hb + 2 * (ks & 1)) : ((2 * ks) | hb);
      wfo[nt][ks] = brow + (unsigned)(row * 128 + ((cid ^ ((row >> 1) & 7)) << 4));
    }
  }
  // fp8: E8M0 scale bytes of this lane's W rows (per n tile) and activation rows (per m tile) of the current output tile
  int wsc[2] = {127, 127}, xsc[4] = {127, 127, 127, 127};
  auto wfrag = [&](int slot, int nt, int ks) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8 *>(gemm_smem + slot * kSlot + wfo[nt][ks]);
  };

  f32x16 acc[4][2];                                                      // [m tile][n tile], transposed: lane -> m, regs -> n
  bf16x8 xf[2][4], wf[2][4];                                             // activation rows (2 m tiles), W rows (2 n tiles) x 4 k steps
  int ct = 0, ck = 0;                                                    // compute cursor
  int cm0, cn0;
  tile_mn(0, cm0, cn0);

  auto load_scales = [&](int m0, int n0) __attribute__((always_inline)) {
    if (DT != 1) return;
    const int a = lane & 31;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int row = S6D_GEMM_QT ? 32 * ((a >> 2) & 1) + 16 * nt + 4 * (a >> 3) + (a & 3) : 32 * nt + a;
      wsc[nt] = (int)p.sw[n0 + wc * 64 + row];
    }
    if (AMX) return;                                                     // the activation scales come per K tile (mx_issue)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xsc[mt] = (int)p.sa[min(m0 + wr * 128 + mt * 32 + a, p.M - 1)];
  };
  // AMX: xsc[] holds the (shifted) scale dwords of the K tile being multiplied, xnx[] those of the next one (in flight or landed);
  // the cursor walks the K tiles of this workgroup's stream like the DMA cursors do
  int xnx[4] = {0, 0, 0, 0};
  int mx_kt = 0, mx_tile = 0;
  unsigned mx_row = 0;                                                   // dword index of (this lane's row of m tile 0, K tile 0) of the cursor's tile
  auto mx_set = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_mn(tile, m0, n0);
    mx_row = (unsigned)(m0 + wr * 128 + (lane & 31)) * (unsigned)p.nk;
  };
  auto mx_issue = [&]() __attribute__((always_inline)) {               // request the cursor's K tile into xnx[], advance the cursor
    if (!AMX) return;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xnx[mt] = (int)p.sa_mx[mx_row + (unsigned)(mt * 32 * p.nk) + (unsigned)mx_kt];
    if (++mx_kt == p.nk) {
      mx_kt = 0;
      if (++mx_tile < my_tiles) mx_set(mx_tile);
    }
  };
  auto mx_take = [&]() __attribute__((always_inline)) {                // xnx[] has landed: it becomes the current K tile's
    if (!AMX) return;
    const int sh = (lane >> 5) * 8;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xsc[mt] = (int)((unsigned)xnx[mt] >> sh);
  };
  // EPI 3 / 4 (folded LayerNorm): rstd of this lane's row in strip mt, applied in the epilogue
  float ln_rs[4] = {1.f, 1.f, 1.f, 1.f};
  // EPI 2: the residual values of the NEXT tile in the accumulator layout -- [strip mt][q]: columns 32 h + 8 q .. + 7 of this lane's
  // row (q = 2 nt + j).  Fetched strip by strip during the current tile's epilogue, right after the strip's stores (its 32
  // accumulator registers are free by then), and first read by init_acc of the next tile, behind the drain that the bias read
  // spends there anyway: the HBM latency of a tile that no other workgroup shares hides behind a whole epilogue.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rn[4][4];
  auto load_resid = [&](int mt, int m0, int n0) __attribute__((always_inline)) {
    const int m = min(m0 + wr * 128 + mt * 32 + (lane & 31), p.M - 1);     // rows past M: a valid address, never stored
    const char *src = (const char *)p.R + (size_t)m * p.ldr2 + (size_t)(n0 + wc * 64 + 32 * (lane >> 5)) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) rn[mt][q] = *reinterpret_cast<const u32x4 *>(src + 16 * q);
  };
  auto init_acc = [&](int m0, int n0) __attribute__((always_inline)) {
    // Accumulators start at the bias.  (`hi ? bs[c1] : bs[c0]` compiles to a per-lane address and eight 16-byte vector loads per
    // tile, waited for with vmcnt(0) -- a drain of the previous epilogue's stores and of the prefetched K tiles at every tile
    // start.  Measured harmless: a build that reads the bias through scalar registers instead, and the no-bias instantiation, run
    // within +-1.5 % of this form on all four ViT-H shapes, profiles/r03_lnfold.txt.)
    // EPI 3 / 4: acc = sigma_m b'_n - mean_m s_n, so that the epilogue's single multiply by rstd_m = 1 / sigma_m gives
    // rstd_m (x W'^T - mean_m s) + b'.  Everything the fold needs is fetched HERE, next to the bias loads' drain: a load in the
    // epilogue would have to wait (in-order counter) for the next tile's LDS-DMA prefetch, still on its way from HBM -- measured
    // 2 us per tile, + 6 % on qkv and lin1.
    const int nb = __builtin_amdgcn_readfirstlane(n0 + wc * 64);
    const S6D_CONST(float) *bs = (const S6D_CONST(float) *)p.bias + nb;
    const S6D_CONST(float) *cs = (const S6D_CONST(float) *)p.CS + nb;
    const bool hi = (lane >> 5) != 0;
    float ln_sig[4] = {0.f, 0.f, 0.f, 0.f}, ln_nmu[4] = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == 3 || EPI == 4)) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = min(m0 + wr * 128 + mt * 32 + (lane & 31), p.M - 1);   // rows past M: a valid address, never stored
        const float2 st = *reinterpret_cast<const float2 *>(p.RS + (size_t)m * 2);   // (mean, sigma = sqrt(var + eps))
        ln_nmu[mt] = -st.x;
        ln_sig[mt] = st.y;
        ln_rs[mt] = __builtin_amdgcn_rcpf(st.y);                             // 1 ulp: far inside the bf16 rounding of the result
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      // EPI 3 / 4: one n tile's 2 x 16 constants at a time (all 64 at once spill loop-invariant registers into the main loop)
      if ((EPI == 3 || EPI == 4) && nt == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // register 4 qd + e of lane half h: column 32 h + 16 nt + 4 qd + e (QT) / 32 nt + 8 qd + 4 h + e
          const int c0 = S6D_GEMM_QT ? 16 * nt + 4 * qd + e : 32 * nt + 8 * qd + e;
          const int c1 = S6D_GEMM_QT ? c0 + 32 : c0 + 4;
          const float b = HAS_BIAS ? (hi ? bs[c1] : bs[c0]) : 0.f;
          if ((EPI == 3 || EPI == 4)) {
            const float sn = hi ? cs[c1] : cs[c0];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt][4 * qd + e] = fmaf(ln_sig[mt], b, ln_nmu[mt] * sn);
          } else if (EPI == 2) {
            // bias + residual, both exact in fp32: x + (a W^T + b) is then rounded to bf16 once, in the epilogue
            static_assert(S6D_GEMM_QT || EPI != 2, "residual registers are laid out for the quad-transposed column order");
            const int r = 4 * qd + e;                                   // column 32 h + 16 nt + r: word (r & 7) >> 1 of piece 2 nt + (r >> 3)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              const unsigned w = rn[mt][2 * nt + (r >> 3)][(r & 7) >> 1];
              acc[mt][nt][r] = b + __uint_as_float((r & 1) ? (w & 0xffff0000u) : (w << 16));
            }
          } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt][4 * qd + e] = b;
          }
        }
      }
    }
  };

  auto epilogue_one = [&](int mt, int nt, int m0, int n0) __attribute__((always_inline)) {   // one 32 x 32 accumulator tile
    const int hb = lane >> 5;
    const int m = m0 + wr * 128 + mt * 32 + (lane & 31);
    unsigned pk[4][2];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
      if (EPI == 1) gelu_erf4(v);
      pk[qd][0] = pack_out<DT>(v[0], v[1]);
      pk[qd][1] = pack_out<DT>(v[2], v[3]);
    }
    // quad qd holds columns 8 qd + 4 hb + {0..3}: swapping the upper lane half of quad 2j with the lower lane half of
    // quad 2j + 1 leaves each lane 8 consecutive columns 16 j + 8 hb + {0..7} of its row
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint4 o;
      {
        auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
        o.x = s0[0];
        o.y = s1[0];
        o.z = s0[1];
        o.w = s1[1];
      }
      if (S6D_GEMM_ABLATE & 4) {
#ifndef HIPEMU
        asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));   // keep the epilogue arithmetic alive
#endif
      } else if (m < p.M) {
        u16 *dst = p.C + (size_t)m * p.ldc + (n0 + wc * 64 + nt * 32 + 16 * j + 8 * hb);
        *reinterpret_cast<uint4 *>(dst) = o;
      }
    }
  };
  // S6D_GEMM_QT: the 32 x 64 strip of m tile mt.  A lane holds 4 chunks of 8 consecutive columns of its row (chunk 2 nt + k =
  // columns 32 h + 16 nt + 8 k ..); the 4 x 4 transpose inside each lane quad turns that into chunk (lane & 3) of the four rows
  // of the quad, i.e. a quad writes 64 contiguous bytes per instruction
  char *ep_base = nullptr;                                               // this tile's store base of the lane and the row stride in bytes
  long ep_row = 0;
  auto epilogue_qt = [&](int mt, int m0, int n0) __attribute__((always_inline)) {
    if (EPI == 5) {
      // GELU, then this lane's 32 columns (32 h + 16 nt + r, r = 0..15: acc[mt][0][*] then acc[mt][1][*]) as one MX block: the
      // quantisation rule of s6d_layernorm_fp8 / utils/fp8.py per BLOCK -- scale 2^e with the smallest e for which amax / 2^e <= 448
      float v[32], amax = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[16 * nt + r + i] = acc[mt][nt][r + i];
          gelu_erf4(v + 16 * nt + r);
#pragma unroll
          for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fabsf(v[16 * nt + r + i]));
        }
      int e2 = 0;
      if (amax > 0.f) {
        int ex;
        const float f = frexpf(amax, &ex);                               // amax = f 2^ex, f in [0.5, 1): 512 f <= 448 iff f <= 0.875
        e2 = ex - (f <= 0.875f ? 9 : 8);
        e2 = min(max(e2, -127), 127);
      }
      const float inv = ldexpf(1.f, -e2);
      unsigned q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i] * inv, v[4 * i + 1] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * inv, v[4 * i + 3] * inv, w, true);
        q[i] = (unsigned)w;
      }
      const int m = m0 + wr * 128 + mt * 32 + (lane & 31);
      const int col = n0 + wc * 64 + 32 * (lane >> 5);
      if (m < p.M) {
        unsigned char *dst = reinterpret_cast<unsigned char *>(p.C) + (size_t)m * p.ldc + col;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(q[0], q[1], q[2], q[3]);
        *reinterpret_cast<uint4 *>(dst + 16) = make_uint4(q[4], q[5], q[6], q[7]);
        p.SC[(size_t)m * (p.N >> 5) + (col >> 5)] = (unsigned char)(e2 + 127);
      }
      return;
    }
    if (EPI == 2 && p.SP) {
      // partial LayerNorm statistics of the row this lane owns over its 32 columns (32 h + {0..31} of the wave's 64): sum and the
      // sum of squared deviations from the group mean, of the fp32 results (the bf16 rounding of the stored values is zero-mean
      // and 2^-9 relative: it moves the mean by ~1e-4 sigma)
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[mt][nt][r];
      const float mean = sum * (1.f / 32.f);
      float m2 = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[mt][nt][r] - mean;
          m2 = fmaf(d, d, m2);
        }
      const int m = m0 + wr * 128 + mt * 32 + (lane & 31);
      const int pidx = ((n0 + wc * 64) >> 5) + (lane >> 5);
      if (m < p.M) {
        p.SP[(size_t)(2 * pidx) * p.M + m] = sum;
        p.SP[(size_t)(2 * pidx + 1) * p.M + m] = m2;
      }
    }
    unsigned X[4][4];                                                    // [chunk][dword]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = acc[mt][nt][8 * k + i];
          if ((EPI == 3 || EPI == 4)) v[i] *= ln_rs[mt];
        }
        if (EPI == 1 || EPI == 4) {
          gelu_erf4(v);
          gelu_erf4(v + 4);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) X[2 * nt + k][d] = pack_out<DT>(v[2 * d], v[2 * d + 1]);
      }
    // bit 0 of (chunk index, lane): pairs (0,1), (2,3) with the neighbour lane ^ 1; then bit 1: pairs (0,2), (1,3) with lane ^ 2
    quad_xchg4<0>(X[0], X[1]);
    quad_xchg4<0>(X[2], X[3]);
    quad_xchg4<1>(X[0], X[2]);
    quad_xchg4<1>(X[1], X[3]);
    // rows 32 mt + (lane & 28) + y of the wave's 128: ep_base (set once per tile by epilogue()) + a constant row stride
    const int mq = m0 + wr * 128 + mt * 32 + (lane & 28);                // first row of this lane's quad
    char *dst = ep_base + (long)(mt * 32) * ep_row;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      if (S6D_GEMM_ABLATE & 4) {
#ifndef HIPEMU
        asm volatile("" ::"v"(X[y][0]), "v"(X[y][1]), "v"(X[y][2]), "v"(X[y][3]));
#endif
      } else if (mq + y < p.M) {
        *reinterpret_cast<uint4 *>(dst + y * ep_row) = make_uint4(X[y][0], X[y][1], X[y][2], X[y][3]);
      }
    }
  };
  auto epilogue = [&](int m0, int n0, int nm0, int nn0) __attribute__((always_inline)) {   // (nm0, nn0): the next tile (EPI 2)
    if (S6D_GEMM_QT && EPI != 5) {
      // store addresses of the tile, once (round 6): element (row, col) lives at C + row ldc + col, or -- column blocks of width
      // cblk stored as separate (M, cblk) matrices -- at C + ((col / cblk) M + row) cblk + col % cblk; either way the rows of one
      // column are a constant stride apart, so a strip's stores are base + (32 mt + y) * stride (was: a division per strip)
      const int row = m0 + wr * 128 + (lane & 28);
      const int col = n0 + wc * 64 + 32 * (lane >> 5) + 8 * (lane & 3);
      if (p.cblk > 0) {
        const int blk = col / p.cblk;
        ep_base = (char *)(p.C + ((size_t)blk * p.M + (size_t)row) * p.cblk + (col - blk * p.cblk));
        ep_row = 2 * (long)p.cblk;
      } else {
        ep_base = (char *)(p.C + (size_t)row * p.ldc + col);
        ep_row = 2 * p.ldc;
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (S6D_GEMM_QT) {
        epilogue_qt(mt, m0, n0);
        if (EPI == 2) load_resid(mt, nm0, nn0);
      } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) epilogue_one(mt, nt, m0, n0);
      }
    }
  };
#define S6D_MFMA(C, A, B)                                                              \
  do {                                                                                 \
    if (S6D_GEMM_ABLATE & 2) {                                                         \
      S6D_PIN(B);                                                                      \
    } else {                                                                           \
      if constexpr (DT == 2) C = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)(A), (f16x8)(B), C, 0, 0, 0); \
      else C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);              \
    }                                                                                  \
  } while (0)
#define S6D_MSEG(QM, QN)                                                               \
  do {                                                                                 \
    S6D_SETPRIO(1);                                                                    \
    S6D_PIN(wf[QN][0]);                                                                \
    S6D_PIN(wf[QN][1]);                                                                \
    S6D_PIN(wf[QN][2]);                                                                \
    S6D_PIN(wf[QN][3]);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                 \
      S6D_MFMA(acc[2 * QM][QN], wf[QN][ks], xf[0][ks]);                                \
      S6D_MFMA(acc[2 * QM + 1][QN], wf[QN][ks], xf[1][ks]);                            \
    }                                                                                  \
    S6D_PIN(acc[2 * QM][QN]);                                                          \
    S6D_PIN(acc[2 * QM + 1][QN]);                                                      \
    S6D_SETPRIO(0);                                                                    \
  } while (0)

  if (S6D_GEMM_PH2) {
    // =============================== two phases per K tile ===============================
    // stream order B0 B1 A0 A1 per K tile, element e in slot e % 10; K tile g: s0 = (4 g) % 10.
    //   phase A(g): reads W rows (both n tiles) of B[wc >> 1](g) and activation rows 0..63 of A[wr](g); issues B0, B1 of K tile g + 2
    //               into the slots A(g - 1) left one phase ago; 16 MFMAs: m tiles 0, 1 x n tiles 0, 1
    //   phase B(g): reads activation rows 64..127; issues A0, A1 of K tile g + 2 into the slots B(g) left one phase ago; counted wait
    //               "K tile g + 1 has landed" (the 4 half-tiles of K tile g + 2 stay in flight); 16 MFMAs: m tiles 2, 3
    // A slot is restaged one phase after its last fragment read; that is safe because every load segment ends with lgkmcnt(0)
    // BEFORE its barrier (the reads are retired when the other wave group, one barrier apart, starts issuing into the slot), and
    // a landed K tile is read one phase (two barriers) after the wait that retired it.
#define S6D_MFMA8(C, A, B, SA, SB, OPB) C = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 0, 0, 0, SA, OPB, SB)
#define S6D_MSEG2_H(QM, H, OPB)                                                        \
  do {                                                                                 \
    S6D_MFMA8(acc[2 * QM][0], wq[0][H], xq[0][H], wsc[0], xsc[2 * QM], OPB);           \
    S6D_MFMA8(acc[2 * QM + 1][0], wq[0][H], xq[1][H], wsc[0], xsc[2 * QM + 1], OPB);   \
    S6D_MFMA8(acc[2 * QM][1], wq[1][H], xq[0][H], wsc[1], xsc[2 * QM], OPB);           \
    S6D_MFMA8(acc[2 * QM + 1][1], wq[1][H], xq[1][H], wsc[1], xsc[2 * QM + 1], OPB);   \
  } while (0)
#define S6D_MSEG2(QM)                                                                  \
  do {                                                                                 \
    S6D_SETPRIO(1);                                                                    \
    if constexpr (DT != 1) {                                                           \
      _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) S6D_PIN(wf[nt][ks]); \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                               \
        S6D_MFMA(acc[2 * QM][0], wf[0][ks], xf[0][ks]);                                \
        S6D_MFMA(acc[2 * QM + 1][0], wf[0][ks], xf[1][ks]);                            \
        S6D_MFMA(acc[2 * QM][1], wf[1][ks], xf[0][ks]);                                \
        S6D_MFMA(acc[2 * QM + 1][1], wf[1][ks], xf[1][ks]);                            \
      }                                                                                \
    } else {                                                                           \
      i32x8 wq[2][2], xq[2][2];                                                        \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int h = 0; h < 2; ++h) { \
        wq[j][h] = __builtin_shufflevector((i32x4)wf[j][2 * h], (i32x4)wf[j][2 * h + 1], 0, 1, 2, 3, 4, 5, 6, 7); \
        xq[j][h] = __builtin_shufflevector((i32x4)xf[j][2 * h], (i32x4)xf[j][2 * h + 1], 0, 1, 2, 3, 4, 5, 6, 7); \
        S6D_PIN(wq[j][h]);                                                             \
      }                                                                                \
      S6D_MSEG2_H(QM, 0, 0);                                                           \
      S6D_MSEG2_H(QM, 1, (AMX ? 2 : 0));                                               \
    }                                                                                  \
    S6D_PIN(acc[2 * QM][0]);                                                           \
    S6D_PIN(acc[2 * QM + 1][0]);                                                       \
    S6D_PIN(acc[2 * QM][1]);                                                           \
    S6D_PIN(acc[2 * QM + 1][1]);                                                       \
    S6D_SETPRIO(0);                                                                    \
  } while (0)
#ifdef HIPEMU
#define S6D_LGKM0()
#else
#define S6D_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
    set_b(0);
    set_a(0);
    if (AMX) {                                                           // scale dwords of K tile 0: older than every DMA piece below
      mx_set(0);
      mx_issue();
    }
    issue_b(0, 0);
    issue_b(1, 1);
    issue_a(0, 2);
    issue_a(1, 3);
    if (G > 1) {
      issue_b(0, 4);
      issue_b(1, 5);
      issue_a(0, 6);
      issue_a(1, 7);
      S6D_VMCNT(8);                                                      // K tile 0 has landed, K tile 1 in flight
    } else {
      S6D_VMCNT(0);
    }
    mx_take();
    if (EPI == 2) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) load_resid(mt, cm0, cn0);
    }
    init_acc(cm0, cn0);
    load_scales(cm0, cn0);
    S6D_BARRIER();
    if (wr == 1) S6D_BARRIER();                                          // the M halves run one barrier apart from here on
    int s0 = 0;
    for (int g = 0; g < G; ++g) {
      const int sA = ring(s0 + 2 + wr), sB = s0 + (wc >> 1);
      const bool more2 = g + 2 < G;
      // ---- phase A
      if (AMX && g + 1 < G) mx_issue();                                  // scale dwords of K tile g + 1, in front of this tile's DMA issues
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[nt][ks] = wfrag(sB, nt, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xf[i][ks] = frag(sA, (unsigned)(i * 4096), ks);
      if (more2) {
        issue_b(0, ring(s0 + 8));                                        // B0, B1 of K tile g + 2 -> slots of A0, A1 (g - 1)
        issue_b(1, ring(s0 + 9));
      }
      S6D_LGKM0();
      S6D_BARRIER();
      S6D_MSEG2(0);
      S6D_BARRIER();
      // ---- phase B
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xf[i][ks] = frag(sA, (unsigned)((2 + i) * 4096), ks);
      if (more2) {
        issue_a(0, s0);                                                  // A0, A1 of K tile g + 2 -> slots of B0, B1 (g)
        issue_a(1, s0 + 1);
        S6D_VMCNT(8);                                                    // K tile g + 1 has landed (K tile g + 2: 4 half-tiles in flight)
      } else {
        S6D_VMCNT(0);
      }
      S6D_LGKM0();
      S6D_BARRIER();
      S6D_MSEG2(1);
      S6D_BARRIER();
      s0 = ring(s0 + 4);
      if (AMX && g + 1 < G) mx_take();                                   // K tile g + 1's scale dwords (landed: older than the counted wait above)
      if (++ck == p.nk) {
        ck = 0;
        if (wr == 0) S6D_BARRIER();
        if (EPI == 2) {
          // unconditional on purpose (after the last tile it re-reads that tile and initialises accumulators nobody uses): a load
          // that is consumed on one path only stays "pending" on the other, and the compiler then drains inside the main loop
          int nm0, nn0;
          tile_mn(min(ct + 1, my_tiles - 1), nm0, nn0);
          epilogue(cm0, cn0, nm0, nn0);
          ++ct;
          cm0 = nm0;
          cn0 = nn0;
          init_acc(cm0, cn0);
        } else {
          epilogue(cm0, cn0, 0, 0);
          if (++ct < my_tiles) {
            tile_mn(ct, cm0, cn0);
            init_acc(cm0, cn0);
            load_scales(cm0, cn0);
          }
        }
        if (wr == 1) S6D_BARRIER();
      }
    }
    if (wr == 0) S6D_BARRIER();
#undef S6D_MSEG2
    return;
  }
  static_assert(DT != 1 || S6D_GEMM_PH2, "the fp8 operands are wired into the two-phase main loop only");
  static_assert(DT == 0 || EPI < 2 || EPI == 5, "the residual K tiles and the folded LayerNorm are wired for bf16 operands");
  static_assert(S6D_GEMM_QT || EPI < 2, "the residual / LayerNorm / MX epilogues extend the quad-transposed epilogue");
  // ---- prologue: half-tiles 0..6 of the stream (K tile 0 whole; B0 B1 A0 of K tile 1)
  set_b(0);
  set_a(0);
  issue_b(0, 0);
  issue_b(1, 1);
  issue_a(0, 2);
  issue_a(1, 3);
  if (G > 1) {
    issue_b(0, 4);
    issue_b(1, 5);
    issue_a(0, 6);
    S6D_VMCNT(6);
  } else {
    S6D_VMCNT(0);
  }
  if (EPI == 2) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) load_resid(mt, cm0, cn0);
  }
  init_acc(cm0, cn0);
  S6D_BARRIER();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wf[0][ks] = wfrag(wc >> 1, 0, ks);        // B(0), n tile 0
  if (wr == 1) S6D_BARRIER();                             // the M halves run one barrier apart from here on

  int s0 = 0;                                                            // ring slot of B0 of K tile g: (4 g) % 10
  for (int g = 0; g < G; ++g) {
    const int sA = ring(s0 + 2 + wr);                                    // A[wr](g)
    const int sB = s0 + (wc >> 1);                                       // B[wc >> 1](g)
    const int sBn = ring(s0 + 4 + (wc >> 1));                            // B[wc >> 1](g + 1)
    const bool more1 = g + 1 < G, more2 = g + 2 < G;

    // phase 1: activation rows 0..63 of this wave -> quadrant (0,0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xf[i][ks] = frag(sA, (unsigned)(i * 4096), ks);
    if (more1) issue_a(1, ring(s0 + 7));                                 // A1(g + 1)
    S6D_BARRIER();
    S6D_MSEG(0, 0);
    S6D_BARRIER();

    // phase 2: W rows 32..63 of this wave -> quadrant (0,1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[1][ks] = wfrag(sB, 1, ks);
    if (more2) issue_b(0, ring(s0 + 8));                                 // B0(g + 2)
    S6D_BARRIER();
    S6D_MSEG(0, 1);
    S6D_BARRIER();

    // phase 3: activation rows 64..127 -> quadrant (1,0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xf[i][ks] = frag(sA, (unsigned)((2 + i) * 4096), ks);
    if (more2) {
      issue_b(1, ring(s0 + 9));                                          // B1(g + 2)
      S6D_VMCNT(8);                                                      // B(g + 1) has landed (4 younger half-tiles in flight)
    } else {
      S6D_VMCNT(0);
    }
    S6D_BARRIER();
    S6D_MSEG(1, 0);
    S6D_BARRIER();

    // phase 4: next K tile's W rows 0..31 (wf[0] is free after quadrant (1,0)) -> quadrant (1,1)
    if (more1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[0][ks] = wfrag(sBn, 0, ks);
    }
    if (more2) {
      issue_a(0, s0);                                                    // A0(g + 2) into the slot B0(g) just left
      S6D_VMCNT(6);                                                      // A(g + 1) has landed (3 younger half-tiles in flight)
    } else {
      S6D_VMCNT(0);
    }
    S6D_BARRIER();
    S6D_MSEG(1, 1);
    S6D_BARRIER();

    s0 = ring(s0 + 4);
    if (++ck == p.nk) {                                                  // output tile complete
      ck = 0;
      // both M halves run the epilogue in the same interval (their VALU work shares the SIMDs either way; running it
      // one after the other would leave one wave per SIMD): the leading half waits one barrier, the trailing half gives
      // one back afterwards, which also restores the one-barrier stagger
      if (wr == 0) S6D_BARRIER();
      if (EPI == 2) {
        int nm0, nn0;
        tile_mn(min(ct + 1, my_tiles - 1), nm0, nn0);
        epilogue(cm0, cn0, nm0, nn0);
        ++ct;
        cm0 = nm0;
        cn0 = nn0;
        init_acc(cm0, cn0);
      } else {
        epilogue(cm0, cn0, 0, 0);
        if (++ct < my_tiles) {
          tile_mn(ct, cm0, cn0);
          init_acc(cm0, cn0);
        }
      }
      if (wr == 1) S6D_BARRIER();
    }
  }
  if (wr == 0) S6D_BARRIER();                             // same barrier count for both halves
#undef S6D_MSEG
#undef S6D_MFMA
}

template <int EPI, bool HAS_BIAS>
__global__ void __launch_bounds__(512, 2) gemm_bf16_kernel(GemmParams p) {
  gemm_body<EPI, HAS_BIAS, 0>(p);
}
template <int EPI, bool HAS_BIAS>
__global__ void __launch_bounds__(512, 2) gemm_fp8_kernel(GemmParams p) {
  gemm_body<EPI, HAS_BIAS, 1>(p);
}
template <int EPI, bool HAS_BIAS>
__global__ void __launch_bounds__(512, 2) gemm_fp8mx_kernel(GemmParams p) {       // activations with MX block scales
  gemm_body<EPI, HAS_BIAS, 1, true>(p);
}
template <int EPI, bool HAS_BIAS>
__global__ void __launch_bounds__(512, 2) gemm_f16_kernel(GemmParams p) {
  gemm_body<EPI, HAS_BIAS, 2>(p);
}


// =====================================================================================================================================
// Version 2: two INDEPENDENT 256-thread workgroups per CU, 256 x 128 output tiles.  In the product this is the path of the shapes the
// 256 x 256 kernel cannot tile -- N % 256 == 128 (gemm_launch: impl 2; plain and GELU epilogues, bf16) -- and nothing else: on the
// ViT-H shapes it measured 0.89 - 0.96 x of version 1 (profiles/r02_gemm_v2_shapes.json) and is not selected there.
//
// What the measurements of version 1 said (profiles/r02_gemm_v1_time.json, 65536 x 1280 -> 5120): the whole kernel 0.76 ms, without
// the epilogue stores 0.65, without the LDS-DMA 0.62, matrix work + fragment reads alone 0.55 -- and everything EXCEPT the matrix work
// alone 0.58.  The eight waves of that kernel are one lock-step machine: nothing runs beside the epilogue of a tile (14 %), and every
// 256-cycle matrix segment is bracketed by workgroup barriers.  Here the CU holds two workgroups that know nothing of each other
// (80 KiB of LDS each, <= 256 VGPRs): on every SIMD one wave of each; whenever one waits -- barrier, DMA landing, fragment reads,
// its GELU / store epilogue -- the other one has the matrix pipe.  Price: a 256 x 128 tile moves 1.5x the operand bytes per FLOP.
//
//   * waves 2 (M) x 2 (N), the same 128 x 64 wave tile, fragment layout, transposed product and epilogue as version 1;
//   * LDS = ring of 5 slots of 16 KiB (128 rows x 64 k, the version-1 image and swizzle); the operand stream is W, A0, A1 per K
//     tile (A0 / A1 = rows 0-127 / 128-255 of the tile) and runs across output tiles;
//   * per K tile: [counted wait + barrier: the K tile has landed] -> ALL fragments of the K tile into registers (24 ds_read_b128,
//     96 VGPRs) -> [barrier: its three slots are free] -> the next three stream elements are issued (12 pieces per wave, between
//     the MFMAs) -> 32 MFMAs from registers.  The stream therefore stays 5 slots = 1.67 K tiles ahead of the matrix work.
// DT (round 6): 0 = bf16, 2 = IEEE half operands and output (as gemm_body's).  Besides N % 256 == 128 this kernel now also serves the
// plain / GELU launches whose 256 x 256 tiling would leave most CUs idle (gemm_launch: fewer than 160 tiles, e.g. the PEM ViT-B's
// 6304 x 768 products: 75 tiles -> 150): half the work per tile, twice the tiles, the same products in the same order per element.
template <int EPI, bool HAS_BIAS, int DT = 0>
__global__ void __launch_bounds__(256, 2) gemm2_bf16_kernel(GemmParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;       // M half / N half of the 256 x 128 tile

  // tile schedule: as version 1 (XCD x takes a contiguous range of the logical order, GM m-tiles per n-tile group)
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int tq = p.ntiles >> 3, tr = p.ntiles & 7;
  const int first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int cnt = tq + (xcd < tr ? 1 : 0);
  if (bi >= cnt) return;
  const int my_tiles = (cnt - bi + bpx - 1) / bpx;
  const int G = my_tiles * p.nk;
  auto tile_mn = [&](int j, int &m0, int &n0) __attribute__((always_inline)) {
    const int L = first + bi + j * bpx;
    const int tpg = p.GM * p.NT;
    const int grp = L / tpg, rem = L - grp * tpg;
    const int mf = grp * p.GM;
    const int gs = min(p.GM, p.MT - mf);
    const int nn = rem / gs;
    m0 = (mf + rem - nn * gs) * 256;
    n0 = nn * 128;
  };

  // staging: wave w fills rows [32 w, 32 w + 32) of a slot in four 1-KiB pieces (8 rows x 128 B); lane -> row (lane >> 3),
  // chunk position (lane & 7), fetched chunk = position ^ ((row >> 1) & 7)
  const int srow = wave * 32 + (lane >> 3);
  unsigned sc[4];                                                       // chunk byte offset of piece i (rows + 8 i)
#pragma unroll
  for (int i = 0; i < 4; ++i) sc[i] = (unsigned)(((lane & 7) ^ (((lane >> 3) + 8 * i) >> 1 & 7)) << 4);
  // issue cursor = the K tile whose W / A0 go out next (K tile in the output tile, tile); its A1 addresses are parked in
  // a1_pend and go out one iteration later (the stream order is W, A0, A1 per K tile, 5 elements ahead of the matrix work)
  unsigned w_off = 0, a_off[2][4], a1_pend[4] = {0u, 0u, 0u, 0u};
  int is_kt = 0, is_tile = 0;
  auto set_tile = [&](int tile) __attribute__((always_inline)) {
    int m0, n0;
    tile_mn(tile, m0, n0);
    w_off = (unsigned)(n0 + srow) * p.ldw2;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) a_off[a][i] = (unsigned)min(m0 + a * 128 + srow + 8 * i, p.M - 1) * p.lda2 + sc[i];
  };
  auto dma = [&](const u16 *base, unsigned off, int slot, int piece) __attribute__((always_inline)) {
    S6D_LDS(char) *dst = (S6D_LDS(char) *)gemm_smem + slot * kSlot + (wave * 4 + piece) * 1024;
    __builtin_amdgcn_global_load_lds((const S6D_GLOBAL(void) *)((const char *)base + off), dst, 16, 0, 0);
  };
  auto issue_w = [&](int slot) __attribute__((always_inline)) {        // W of the cursor's K tile
    const unsigned k = (unsigned)is_kt * 128u;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(p.W, w_off + (unsigned)(8 * i) * p.ldw2 + sc[i] + k, slot, i);
  };
  auto issue_a0 = [&](int slot) __attribute__((always_inline)) {       // A0 of the cursor's K tile; parks its A1; advances the cursor
    const unsigned k = (unsigned)is_kt * 128u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma(p.A, a_off[0][i] + k, slot, i);
      a1_pend[i] = a_off[1][i] + k;
    }
    if (++is_kt == p.nk) {
      is_kt = 0;
      if (++is_tile < my_tiles) set_tile(is_tile);
    }
  };
  auto issue_a1 = [&](int slot) __attribute__((always_inline)) {       // the parked A1
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(p.A, a1_pend[i], slot, i);
  };

  unsigned foff[4];
  {
    const int sw = (lane >> 1) & 7, hb = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = (unsigned)((lane & 31) * 128 + ((((2 * ks) | hb) ^ sw) << 4));
  }
  auto frag = [&](int slot, unsigned rowbytes, int ks) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8 *>(gemm_smem + slot * kSlot + rowbytes + foff[ks]);
  };

  f32x16 acc[4][2];
  int ct = 0, ck = 0, cm0, cn0;
  tile_mn(0, cm0, cn0);

  // EPI 2 (round 6): C = A W^T + bias + R summed in fp32 as in gemm_body -- the accumulators start at bias + residual (this lane's
  // row, its columns 32 nt + 8 qd + 4 h + e: 8 bytes per (mt, nt, qd), requested at the tile start) -- and the epilogue leaves the
  // same partial row statistics; see epilogue() for how the same BITS come out of another register layout.
  auto init_acc = [&](int m0, int n0) __attribute__((always_inline)) {
    const int nb = __builtin_amdgcn_readfirstlane(n0 + wc * 64);
    const S6D_CONST(float) *bs = (const S6D_CONST(float) *)p.bias + nb;
    const bool hi = (lane >> 5) != 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = HAS_BIAS ? bs[nt * 32 + 8 * qd + e] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float b = hi ? v[4 + e] : v[e];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt][nt][4 * qd + e] = b;
        }
      }
    if (EPI == 2) {
      const int hb = lane >> 5;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = min(m0 + wr * 128 + mt * 32 + (lane & 31), p.M - 1);   // rows past M: a valid address, never stored
        const char *src = (const char *)p.R + (size_t)m * p.ldr2 + (size_t)(n0 + wc * 64 + 4 * hb) * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const uint2 w = *reinterpret_cast<const uint2 *>(src + (32 * nt + 8 * qd) * 2);
            acc[mt][nt][4 * qd] += __uint_as_float(w.x << 16);
            acc[mt][nt][4 * qd + 1] += __uint_as_float(w.x & 0xffff0000u);
            acc[mt][nt][4 * qd + 2] += __uint_as_float(w.y << 16);
            acc[mt][nt][4 * qd + 3] += __uint_as_float(w.y & 0xffff0000u);
          }
      }
    }
  };
  auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
    const int hb = lane >> 5;
    if constexpr (EPI == 2) {
      // The eight-wave kernel feeds the W rows in an order that leaves a lane with the 32 CONSECUTIVE columns 32 h .. 32 h + 31 of
      // its row, and sums its row statistics over them in ascending order.  Here lane half h holds columns 8 qd + 4 h + e of BOTH
      // 32-column tiles; one v_permlane32_swap per register pair hands the lower half tile 0 and the upper half tile 1 complete:
      // lo[r] = columns 8 qd + e, hi[r] = columns 8 qd + 4 + e of tile h (r = 4 qd + e).  Summed qd by qd -- lo, then hi -- that is
      // the same ascending order over the same fp32 values: the same statistics, bit for bit, and the same rounded outputs.
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wr * 128 + mt * 32 + (lane & 31);
        float lo[16], hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[mt][0][r]), __float_as_uint(acc[mt][1][r]), false, false);
          lo[r] = __uint_as_float(sw[0]);
          hv[r] = __uint_as_float(sw[1]);
        }
        if (p.SP) {
          float sum = 0.f;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += lo[4 * qd + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += hv[4 * qd + e];
          }
          const float mean = sum * (1.f / 32.f);
          float m2 = 0.f;
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = lo[4 * qd + e] - mean;
              m2 = fmaf(d, d, m2);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = hv[4 * qd + e] - mean;
              m2 = fmaf(d, d, m2);
            }
          }
          const int pidx = ((n0 + wc * 64) >> 5) + hb;
          if (m < p.M) {
            p.SP[(size_t)(2 * pidx) * p.M + m] = sum;
            p.SP[(size_t)(2 * pidx + 1) * p.M + m] = m2;
          }
        }
        if (m < p.M) {
          u16 *dst = p.C + (size_t)m * p.ldc + (n0 + wc * 64 + 32 * hb);
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            uint4 o;
            o.x = pack_out<DT>(lo[4 * qd], lo[4 * qd + 1]);
            o.y = pack_out<DT>(lo[4 * qd + 2], lo[4 * qd + 3]);
            o.z = pack_out<DT>(hv[4 * qd], hv[4 * qd + 1]);
            o.w = pack_out<DT>(hv[4 * qd + 2], hv[4 * qd + 3]);
            *reinterpret_cast<uint4 *>(dst + 8 * qd) = o;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wr * 128 + mt * 32 + (lane & 31);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        unsigned pk[4][2];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * qd + e];
          if (EPI == 1) gelu_erf4(v);
          pk[qd][0] = pack_out<DT>(v[0], v[1]);
          pk[qd][1] = pack_out<DT>(v[2], v[3]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint4 o;
          auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
          o.x = s0[0];
          o.y = s1[0];
          o.z = s0[1];
          o.w = s1[1];
          if (m < p.M) {
            u16 *dst = p.C + (size_t)m * p.ldc + (n0 + wc * 64 + nt * 32 + 16 * j + 8 * hb);
            *reinterpret_cast<uint4 *>(dst) = o;
          }
        }
      }
    }
  };

  // ---- prologue: the first 5 stream elements (K tile 0 whole; W, A0 of K tile 1)
  set_tile(0);
  issue_w(0);
  issue_a0(1);
  issue_a1(2);
  if (G > 1) {
    issue_w(3);
    issue_a0(4);
  }
  init_acc(cm0, cn0);

  int s0 = 0;                                                            // ring slot of W of K tile g: (3 g) % 5
  for (int g = 0; g < G; ++g) {
    // the K tile has landed: mine by the counted wait (two younger elements = 8 pieces stay in flight), everybody's by the barrier
    if (g + 1 < G) S6D_VMCNT(8); else S6D_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    const int sW = s0, sA = ring5(s0 + 1 + wr);
    bf16x8 wf[2][4], xf[4][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[nt][ks] = frag(sW, (unsigned)((wc * 64 + nt * 32) * 128), ks);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xf[mt][ks] = frag(sA, (unsigned)(mt * 32 * 128), ks);
#ifndef HIPEMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the fragments are in registers ...
#endif
    __builtin_amdgcn_s_barrier();                                        // ... everybody's: the three slots of this K tile are free
    // the next three stream elements go out between the matrix instructions of the four k steps
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks == 0 && g + 1 < G) issue_a1(s0);                            // A1(g + 1) into the slot W(g) left
      if (ks == 1 && g + 2 < G) issue_w(ring5(s0 + 1));                  // W(g + 2) into A0(g)'s
      if (ks == 2 && g + 2 < G) issue_a0(ring5(s0 + 2));                 // A0(g + 2) into A1(g)'s
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          if constexpr (DT == 2)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)(wf[nt][ks]), (f16x8)(xf[mt][ks]), acc[mt][nt], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][ks], xf[mt][ks], acc[mt][nt], 0, 0, 0);
    }
    s0 = ring5(s0 + 3);
    if (++ck == p.nk) {
      ck = 0;
      epilogue(cm0, cn0);
      if (++ct < my_tiles) {
        tile_mn(ct, cm0, cn0);
        init_acc(cm0, cn0);
      }
    }
  }
}

}  // namespace s6d

using namespace s6d;

// Which kernel: the eight-wave 256 x 256 machine wherever N % 256 == 0 (every ViT shape: measured ahead of version 2 on all of them,
// profiles/r02_gemm_v2_shapes.json); version 2 (two independent 256 x 128 workgroups per CU) serves N % 256 == 128.
extern "C" int s6d_gemm_bf16_cblk(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M,
                                  int N, int K, int epilogue, int col_block, int max_blocks, void *stream);
struct GemmExtra {            // operands of the residual (EPI 2) and folded-LayerNorm (EPI 3 / 4) forms
  const void *R = nullptr;    // residual, row stride ldr elements
  long ldr = 0;
  float *SP = nullptr;        // partial row statistics out (optional)
  const float *RS = nullptr;  // row (mean, rstd) in
  const float *CS = nullptr;  // column sums of the folded weight
  unsigned char *SC = nullptr;        // EPI 5: MX scale bytes out, [M][N / 32]
  const unsigned *sa_mx = nullptr;    // MX scale bytes of the A operand in, [M][K / 32]
};
static int gemm_launch(const void *A, long lda, const void *W, long ldw, const float *bias, const GemmExtra &x, void *C, long ldc,
                       int M, int N, int K, int epilogue, int col_block, int max_blocks, void *stream, int dt = 0,
                       const unsigned char *sa = nullptr, const unsigned char *sw = nullptr);

extern "C" int s6d_gemm_f16(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M, int N, int K,
                            int epilogue, int max_blocks, void *stream) {
  if (epilogue != 0 && epilogue != 1) return S6D_EINVAL;
  if (N % 256 != 0 || !S6D_GEMM_QT) return S6D_EUNSUPPORTED;            // the 256 x 256-tile kernel only
  return gemm_launch(A, lda, W, ldw, bias, GemmExtra(), C, ldc, M, N, K, epilogue, 0, max_blocks, stream, 2);
}

extern "C" int s6d_gemm_fp8(const void *A, long lda, const unsigned char *a_scale, const void *W, long ldw, const unsigned char *w_scale,
                            const float *bias, void *C, long ldc, int M, int N, int K, int epilogue, int max_blocks, void *stream) {
  if (!a_scale || !w_scale) return S6D_EINVAL;
  if (epilogue != 0 && epilogue != 1) return S6D_EINVAL;
  if (N % 256 != 0 || K % 128 != 0 || !S6D_GEMM_QT || !S6D_GEMM_PH2) return S6D_EUNSUPPORTED;
  return gemm_launch(A, lda, W, ldw, bias, GemmExtra(), C, ldc, M, N, K, epilogue, 0, max_blocks, stream, 1, a_scale, w_scale);
}

// lin1 of the fp8 block: C8 = e4m3(GELU(A W^T + bias)) (M, N) bytes, row stride ldc bytes, + one E8M0 byte per row and 32 columns at
// c_scale [M][N / 32] -- the MX A operand of s6d_gemm_fp8_mxa.
extern "C" int s6d_gemm_fp8_gelu_mx(const void *A, long lda, const unsigned char *a_scale, const void *W, long ldw,
                                    const unsigned char *w_scale, const float *bias, void *C8, long ldc, unsigned char *c_scale,
                                    int M, int N, int K, int max_blocks, void *stream) {
  if (!a_scale || !w_scale || !c_scale) return S6D_EINVAL;
  if (N % 256 != 0 || K % 128 != 0 || !S6D_GEMM_QT || !S6D_GEMM_PH2) return S6D_EUNSUPPORTED;
  if (ldc < N || (ldc % 16) != 0) return S6D_EINVAL;
  GemmExtra x;
  x.SC = c_scale;
  // (gemm_launch checks ldc against N in ELEMENTS of a 2-byte output: the byte rows of this form satisfy the same inequalities)
  return gemm_launch(A, lda, W, ldw, bias, x, C8, ldc, M, N, K, 5, 0, max_blocks, stream, 1, a_scale, w_scale);
}

// lin2 of the fp8 block: the activations carry MX block scales (a_mx [M][K / 32] E8M0 bytes, what s6d_gemm_fp8_gelu_mx wrote),
// the weights one scale per output channel; bf16 output, bias / GELU epilogue as s6d_gemm_fp8.  The scale dwords of a whole 256-row
// tile are read: a_mx must be READABLE for ceil(M / 256) * 256 rows (rows past M may hold anything; their products are not stored).
extern "C" int s6d_gemm_fp8_mxa(const void *A, long lda, const unsigned char *a_mx, const void *W, long ldw, const unsigned char *w_scale,
                                const float *bias, void *C, long ldc, int M, int N, int K, int epilogue, int max_blocks, void *stream) {
  if (!a_mx || !w_scale || ((uintptr_t)a_mx & 3)) return S6D_EINVAL;
  if (epilogue != 0 && epilogue != 1) return S6D_EINVAL;
  if (N % 256 != 0 || K % 128 != 0 || !S6D_GEMM_QT || !S6D_GEMM_PH2) return S6D_EUNSUPPORTED;
  GemmExtra x;
  x.sa_mx = reinterpret_cast<const unsigned *>(a_mx);
  return gemm_launch(A, lda, W, ldw, bias, x, C, ldc, M, N, K, epilogue, 0, max_blocks, stream, 1, nullptr, w_scale);
}

extern "C" int s6d_gemm_bf16_res(const void *A, long lda, const void *W, long ldw, const float *bias, const void *R, long ldr,
                                 float *stats_partial, void *C, long ldc, int M, int N, int K, int max_blocks, void *stream) {
  if (!R || ldr < N || (ldr % 8) || ((uintptr_t)R & 15)) return S6D_EINVAL;
  if (N % 256 != 0 || !S6D_GEMM_QT) return S6D_EUNSUPPORTED;            // the quad-transposed epilogue of the 256 x 256 kernel
  GemmExtra x;
  x.R = R;
  x.ldr = ldr;
  x.SP = stats_partial;
  return gemm_launch(A, lda, W, ldw, bias, x, C, ldc, M, N, K, 2, 0, max_blocks, stream);
}

extern "C" int s6d_gemm_bf16_lnfold(const void *A, long lda, const float *row_stats, const void *W, long ldw, const float *col_sums,
                                    const float *bias, void *C, long ldc, int M, int N, int K, int gelu, int col_block,
                                    int max_blocks, void *stream) {
  if (!row_stats || !col_sums || !bias) return S6D_EINVAL;
  if (gelu != 0 && gelu != 1) return S6D_EINVAL;
  if (N % 256 != 0 || !S6D_GEMM_QT) return S6D_EUNSUPPORTED;
  GemmExtra x;
  x.RS = row_stats;
  x.CS = col_sums;
  return gemm_launch(A, lda, W, ldw, bias, x, C, ldc, M, N, K, 3 + gelu, col_block, max_blocks, stream);
}

extern "C" int s6d_gemm_bf16(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M,
                             int N, int K, int epilogue, int max_blocks, void *stream) {
  return s6d_gemm_bf16_cblk(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, 0, max_blocks, stream);
}

extern "C" int s6d_gemm_bf16_cblk(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M,
                                  int N, int K, int epilogue, int col_block, int max_blocks, void *stream) {
  if (epilogue != 0 && epilogue != 1) return S6D_EINVAL;
  return gemm_launch(A, lda, W, ldw, bias, GemmExtra(), C, ldc, M, N, K, epilogue, col_block, max_blocks, stream);
}

static int gemm_launch(const void *A, long lda, const void *W, long ldw, const float *bias, const GemmExtra &x, void *C, long ldc,
                       int M, int N, int K, int epilogue, int col_block, int max_blocks, void *stream, int dt,
                       const unsigned char *sa, const unsigned char *sw) {
  if (M < 0 || N <= 0 || K <= 0) return S6D_EINVAL;
  const int esz = dt == 1 ? 1 : 2, kstep = dt == 1 ? 128 : 64, ralign = 16 / esz;   // operand element bytes, K step, elements per 16 bytes
  if (M == 0) return S6D_OK;                                            // an empty row batch: nothing to launch
  if (!A || !W || !C) return S6D_EINVAL;
  if (col_block < 0 || (col_block % 8) != 0 || (col_block > 0 && N % col_block != 0)) return S6D_EINVAL;
  if (col_block > 0 && (N % 256 != 0 || !S6D_GEMM_QT)) return S6D_EUNSUPPORTED;   // the quad-transposed epilogue of the 256 x 256 kernel
  if (N % 128 != 0 || K % kstep != 0 || lda < K || ldw < K || ldc < N) return S6D_EINVAL;
  if ((lda % ralign) || (ldw % ralign) || (ldc % 8)) return S6D_EINVAL;  // 16-byte rows
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) return S6D_EINVAL;
  if (epilogue < 0 || epilogue > 5 || (epilogue == 5 && dt != 1)) return S6D_EINVAL;
  // staging addresses are 32-bit byte offsets from A / W
  if ((double)M * (double)lda * esz >= 2147483648.0 || (double)N * (double)ldw * esz >= 2147483648.0) return S6D_EUNSUPPORTED;
  int impl = (N % 256 != 0) ? 2 : 1;
  // plain / GELU launches that would put fewer than 160 of the 256 x 256 tiles on the 256 CUs take the 256 x 128 kernel (round 6)
  if (impl == 1 && g_s6d_gemm_small_tile && col_block == 0 && !x.sa_mx && (long)((M + 255) / 256) * (N / 256) < 160 &&
      (((dt == 0 || dt == 2) && (epilogue == 0 || epilogue == 1)) || (dt == 0 && epilogue == 2 && g_s6d_gemm_small_tile == 2)))
    impl = 2;     // (epilogue 2 = + residual + row statistics, the ViT-H's proj / lin2 at one frame: built, bit-equal, and SLOWER --
                  // 10.73 against 10.11 ms for the encoder on one frame, a lone 4-wave workgroup per CU has nobody to hide its
                  // barriers -- so only on request: s6d_set_gemm_small_tile(2))
  if (impl == 2 && epilogue > 2) return S6D_EUNSUPPORTED;
  if (impl == 2 && dt != 0 && dt != 2) return S6D_EUNSUPPORTED;
  GemmParams p;
  p.A = (const u16 *)A;
  p.W = (const u16 *)W;
  p.bias = bias;
  p.C = (u16 *)C;
  p.R = (const u16 *)x.R;
  p.ldr2 = (unsigned)(x.ldr * 2);
  p.SP = x.SP;
  p.RS = x.RS;
  p.CS = x.CS;
  p.lda2 = (unsigned)(lda * esz);
  p.ldw2 = (unsigned)(ldw * esz);
  p.sa = sa;
  p.sw = sw;
  p.sa_mx = x.sa_mx;
  p.SC = x.SC;
  p.ldc = ldc;
  p.M = M;
  p.N = N;
  p.K = K;
  p.MT = (M + 255) / 256;
  p.NT = N / (impl == 2 ? 128 : 256);
  p.nk = K / kstep;
  p.ntiles = p.MT * p.NT;
  p.GM = 8;                                                              // m-tiles per n-tile group of the schedule (profiles/r04_gemm_gm.txt)
  p.cblk = col_block;
  hipStream_t st = as_stream(stream);
  // the four-wave form (csrc/s6d_gemm4.hip) where it applies and is selected (s6d_set_gemm_wave_tile; 0 = gemm4_default below)
  if (impl == 1 && g_s6d_gemm_wave_tile != 64 && !x.sa_mx && gemm4_supports(p, epilogue, dt) &&
      (g_s6d_gemm_wave_tile == 128 || gemm4_default(p, epilogue, dt)))
    return gemm4_launch(p, epilogue, max_blocks, st, dt);
  if (impl == 2) {
    if (max_blocks <= 0) max_blocks = 512;                               // two persistent workgroups per CU
    int grid = p.ntiles < max_blocks ? p.ntiles : max_blocks;
    grid = (grid + 7) & ~7;
    const size_t lds = (size_t)5 * kSlot;
#define S6D_GEMM2_LAUNCH(E, HB, D)                                                                                      \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm2_bf16_kernel<E, HB, D>),                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
    hipLaunchKernelGGL((gemm2_bf16_kernel<E, HB, D>), dim3(grid), dim3(256), lds, st, p);                               \
  } while (0)
    if (dt == 2) {
      if (epilogue == 1) {
        if (bias) S6D_GEMM2_LAUNCH(1, true, 2); else S6D_GEMM2_LAUNCH(1, false, 2);
      } else {
        if (bias) S6D_GEMM2_LAUNCH(0, true, 2); else S6D_GEMM2_LAUNCH(0, false, 2);
      }
    } else if (epilogue == 2) {
      if (bias) S6D_GEMM2_LAUNCH(2, true, 0); else S6D_GEMM2_LAUNCH(2, false, 0);
    } else if (epilogue == 1) {
      if (bias) S6D_GEMM2_LAUNCH(1, true, 0); else S6D_GEMM2_LAUNCH(1, false, 0);
    } else {
      if (bias) S6D_GEMM2_LAUNCH(0, true, 0); else S6D_GEMM2_LAUNCH(0, false, 0);
    }
#undef S6D_GEMM2_LAUNCH
    return launch_status();
  }
  if (max_blocks <= 0) max_blocks = 256;                                 // one persistent workgroup per CU
  int grid = p.ntiles < max_blocks ? p.ntiles : max_blocks;
  grid = (grid + 7) & ~7;                                                // whole XCD rounds
  const size_t lds = (size_t)kRing * kSlot;
#define S6D_GEMM_LAUNCH(E, HB)                                                                                          \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_kernel<E, HB>),                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
    hipLaunchKernelGGL((gemm_bf16_kernel<E, HB>), dim3(grid), dim3(512), lds, st, p);                                   \
  } while (0)
#define S6D_GEMM8_LAUNCH(E, HB)                                                                                         \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_fp8_kernel<E, HB>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
    hipLaunchKernelGGL((gemm_fp8_kernel<E, HB>), dim3(grid), dim3(512), lds, st, p);                                    \
  } while (0)
#define S6D_GEMM8X_LAUNCH(E, HB)                                                                                        \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_fp8mx_kernel<E, HB>),                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
    hipLaunchKernelGGL((gemm_fp8mx_kernel<E, HB>), dim3(grid), dim3(512), lds, st, p);                                  \
  } while (0)
  if (dt == 1 && x.sa_mx) {
    if (epilogue == 1) {
      if (bias) S6D_GEMM8X_LAUNCH(1, true); else S6D_GEMM8X_LAUNCH(1, false);
    } else {
      if (bias) S6D_GEMM8X_LAUNCH(0, true); else S6D_GEMM8X_LAUNCH(0, false);
    }
    return launch_status();
  }
#undef S6D_GEMM8X_LAUNCH
  if (dt == 1) {
    if (epilogue == 5) {
      if (bias) S6D_GEMM8_LAUNCH(5, true); else S6D_GEMM8_LAUNCH(5, false);
    } else if (epilogue == 1) {
      if (bias) S6D_GEMM8_LAUNCH(1, true); else S6D_GEMM8_LAUNCH(1, false);
    } else {
      if (bias) S6D_GEMM8_LAUNCH(0, true); else S6D_GEMM8_LAUNCH(0, false);
    }
    return launch_status();
  }
#define S6D_GEMMH_LAUNCH(E, HB)                                                                                         \
  do {                                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f16_kernel<E, HB>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
    hipLaunchKernelGGL((gemm_f16_kernel<E, HB>), dim3(grid), dim3(512), lds, st, p);                                    \
  } while (0)
  if (dt == 2) {
    if (epilogue == 1) {
      if (bias) S6D_GEMMH_LAUNCH(1, true); else S6D_GEMMH_LAUNCH(1, false);
    } else {
      if (bias) S6D_GEMMH_LAUNCH(0, true); else S6D_GEMMH_LAUNCH(0, false);
    }
    return launch_status();
  }
#undef S6D_GEMMH_LAUNCH
#undef S6D_GEMM8_LAUNCH
  if (epilogue == 4) {
    S6D_GEMM_LAUNCH(4, true);
  } else if (epilogue == 3) {
    S6D_GEMM_LAUNCH(3, true);
  } else if (epilogue == 2) {
    if (bias) S6D_GEMM_LAUNCH(2, true); else S6D_GEMM_LAUNCH(2, false);
  } else if (epilogue == 1) {
    if (bias) S6D_GEMM_LAUNCH(1, true); else S6D_GEMM_LAUNCH(1, false);
  } else {
    if (bias) S6D_GEMM_LAUNCH(0, true); else S6D_GEMM_LAUNCH(0, false);
  }
#undef S6D_GEMM_LAUNCH
  return launch_status();
}
