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

// csrc/s6d_gemm4.hip: the four-wave kernel (bf16 operands, EPI 0 - 4).  Returns S6D_EUNSUPPORTED for what it does not cover.
int gemm4_launch(const GemmParams &p, int epilogue, int max_blocks, hipStream_t st, int dt);
bool gemm4_supports(const GemmParams &p, int epilogue, int dt);
bool gemm4_default(const GemmParams &p, int epilogue, int dt);   // the library's own choice (measured per shape class)
extern int g_s6d_gemm_wave_tile;             // csrc/s6d_capi.hip: s6d_set_gemm_wave_tile (0 = by shape, 64 / 128 = forced)

}  // namespace s6d
