// Operand block of the GEMM kernels (csrc/s6d_gemm.hip: the eight-wave 256 x 256 machine and the 256 x 128 form;
// csrc/s6d_gemm4.hip: the four-wave form with 128 x 128 wave tiles).  One definition: gemm_launch fills it once and hands it to
// whichever kernel the shape selects.
#pragma once
#include "s6d_common.h"

namespace s6d {

typedef unsigned short u16;

struct GemmParams {
  const u16 *A;       // (M,K) bf16, row stride lda
  const u16 *W;       // (N,K) bf16, row stride ldw
  const float *bias;  // (N) f32 or nullptr
  u16 *C;             // (M,N) bf16, row stride ldc
  const unsigned char *sa, *sw;   // fp8 operands (DT = 1): E8M0 scale byte of every A row (M) / W row (N); value = q * 2^(byte - 127)
  const unsigned *sa_mx;          // MX form of the A operand (AMX; round 4): one E8M0 byte per row and 32-k block, [M][K / 32] bytes =
                                  // [M][nk] dwords (a dword = the four blocks of one 128-byte K tile); sa is unused then
  unsigned char *SC;              // EPI 5: the output is e4m3 bytes at C (row stride ldc BYTES) + one E8M0 byte per row and 32 columns
                                  // here, [M][N / 32] -- the MX A operand of the next GEMM
  const u16 *R;       // EPI 2: residual (M,N) bf16, row stride ldr2 BYTES; may be C itself (a tile's residual is read by the workgroup
                      // that stores the tile, one tile ahead of its stores)
  float *SP;          // EPI 2, optional: partial row statistics, [N / 32][2][M] floats (sum, sum of squared deviations per 32 columns)
  const float *RS;    // EPI 3 / 4: per-row (mean, sigma = sqrt(var + eps)) of A, [M][2] floats
  const float *CS;    // EPI 3 / 4: s_n = sum_k W'_nk, (N) floats
  unsigned ldr2;
  unsigned lda2, ldw2;  // row strides in BYTES
  long ldc;
  int M, N, K;
  int MT, NT, nk, ntiles;
  int GM;             // m-tiles per group of the tile order
  int cblk;           // 0: C is (M, N) with row stride ldc.  > 0 (multiple of 8): column blocks of this width are stored as
                      // separate (M, cblk) matrices one after the other -- element (m, n) at C + (n / cblk) M cblk + m cblk + n % cblk
                      // (the qkv projection writes q / k / v head-major for the attention kernels: every head's rows contiguous)
};

// One stage of the 4 x 4 dword transpose inside every lane quad (the quad-transposed epilogue of both kernel forms), for four
// register pairs at once: lane bit BIT (0: partner = lane ^ 1, quad_perm [1,0,3,2]; 1: partner = lane ^ 2, quad_perm [2,3,0,1])
// decides which element of a pair (a, b) a lane keeps -- lanes with the bit clear keep a and take the partner's a into b, lanes with
// the bit set keep b and take the partner's b into a.  The exchange and the select are ONE instruction (v_cndmask_b32 with a DPP
// source: D = vcc ? src1 : dpp(src0)); vcc is the "bit clear" lane mask for the a outputs and its complement for the b outputs:
// 8 vector + 3 scalar instructions per four pairs (left to the compiler: select what to send, v_mov_dpp, two selects = 16).
// s_nop 1: the two wait states between a vector write of a register and a DPP read of it (nothing pads an asm statement).
#ifndef HIPEMU
#define S6D_QX_BODY(QP)                                                                                        \
  asm volatile("s_nop 1\n\ts_mov_b64 vcc, %16\n\t"                                                            \
               "v_cndmask_b32_dpp %0, %12, %8, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                       \
               "v_cndmask_b32_dpp %1, %13, %9, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                       \
               "v_cndmask_b32_dpp %2, %14, %10, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                      \
               "v_cndmask_b32_dpp %3, %15, %11, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                      \
               "s_not_b64 vcc, vcc\n\t"                                                                        \
               "v_cndmask_b32_dpp %4, %8, %12, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                       \
               "v_cndmask_b32_dpp %5, %9, %13, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                       \
               "v_cndmask_b32_dpp %6, %10, %14, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                      \
               "v_cndmask_b32_dpp %7, %11, %15, vcc " QP " row_mask:0xf bank_mask:0xf"                           \
               : "=&v"(na[0]), "=&v"(na[1]), "=&v"(na[2]), "=&v"(na[3]), "=&v"(nb[0]), "=&v"(nb[1]), "=&v"(nb[2]), "=&v"(nb[3]) \
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "s"(clear_mask)     \
               : "vcc", "scc")
template <int BIT>
__device__ __forceinline__ void quad_xchg4(unsigned (&a)[4], unsigned (&b)[4]) {
  const unsigned long long clear_mask = BIT == 0 ? 0x5555555555555555ull : 0x3333333333333333ull;
  unsigned na[4], nb[4];
  if constexpr (BIT == 0) S6D_QX_BODY("quad_perm:[1,0,3,2]");
  else S6D_QX_BODY("quad_perm:[2,3,0,1]");
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = na[i], b[i] = nb[i];
}
#undef S6D_QX_BODY
#else   // the host emulator: the same exchange, lane by lane
template <int BIT>
__device__ __forceinline__ void quad_xchg4(unsigned (&a)[4], unsigned (&b)[4]) {
  const bool set = ((threadIdx.x >> BIT) & 1) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned send = set ? a[i] : b[i];
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, BIT == 0 ? 0xB1 : 0x4E, 0xF, 0xF, true);
    a[i] = set ? recv : a[i];
    b[i] = set ? b[i] : recv;
  }
}
#endif

// csrc/s6d_gemm4.hip: the four-wave kernel (bf16 operands, EPI 0 - 4).  Returns S6D_EUNSUPPORTED for what it does not cover.
int gemm4_launch(const GemmParams &p, int epilogue, int max_blocks, hipStream_t st, int dt);
bool gemm4_supports(const GemmParams &p, int epilogue, int dt);
bool gemm4_default(const GemmParams &p, int epilogue, int dt);   // the library's own choice (measured per shape class)
extern int g_s6d_gemm_wave_tile;             // csrc/s6d_capi.hip: s6d_set_gemm_wave_tile (0 = by shape, 64 / 128 = forced)
extern int g_s6d_gemm_small_tile;            // csrc/s6d_capi.hip: s6d_set_gemm_small_tile (1 = 256 x 128 tiles for under-filled plain / GELU launches)

}  // namespace s6d
