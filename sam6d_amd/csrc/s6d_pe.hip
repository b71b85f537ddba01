// PositionalEncoding of the fine point matching, fused (gfx950).
//
// Reference: PositionalEncoding.forward, Pose_Estimation_Model/model/fine_point_matching.py:101-125:
//   QueryAndGroup (ball query + 2 x group_points, pointnet2_utils.py:294-376) -> (B,6,N,ns)
//   SharedMLP [6,32,64,128] = 3 x (1x1 conv + BatchNorm + ReLU) (pytorch_utils.py:25-50) -> (B,128,N,ns): 67 MB/instance
//   max over the ns neighbours.
// Here one wavefront walks a point's neighbourhood in tiles of 16 neighbours and keeps the whole 6->32->64->128
// chain in registers: layer l's output tile in MFMA C layout (feature = g*4+r, neighbour = lane&15) IS layer
// l+1's B operand (k index = feature), with the weight matrix as the A operand (v_mfma_f32_16x16x4_f32, exact
// fp32).  BatchNorm (eval) is folded into the weights by the caller; bias + ReLU commute with the neighbour
// max, so the last layer only keeps a running max of raw accumulators.  Nothing but the (B,N,128) result
// reaches HBM.
#include "s6d_common.h"

namespace s6d {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int PE_WAVES = 4;

__global__ __launch_bounds__(PE_WAVES * 64) void pe_group_mlp_kernel(const float *__restrict__ pts, const int32_t *__restrict__ idx,
                                                                   int B, int N, int ns, const float *__restrict__ W0,
                                                                   const float *__restrict__ b0, const float *__restrict__ W1,
                                                                   const float *__restrict__ b1, const float *__restrict__ W2,
                                                                   const float *__restrict__ b2, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sW1[64 * 32];
  __shared__ __attribute__((aligned(16))) float sW2[128 * 64];
  __shared__ float sW0[32 * 6], sb0[32], sb1[64], sb2[128];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < 64 * 32; i += PE_WAVES * 64) sW1[i] = W1[i];
  for (int i = tid; i < 128 * 64; i += PE_WAVES * 64) sW2[i] = W2[i];
  for (int i = tid; i < 32 * 6; i += PE_WAVES * 64) sW0[i] = W0[i];
  if (tid < 32) sb0[tid] = b0[tid];
  if (tid < 64) sb1[tid] = b1[tid];
  if (tid < 128) sb2[tid] = b2[tid];
  __syncthreads();

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
      const int k = min(tt * 16 + c, ns - 1);                      // surplus lanes repeat the last neighbour (max unchanged)
      const int j = idx[pt * ns + k];
      const float nx = P[j * 3], ny = P[j * 3 + 1], nz = P[j * 3 + 2];
      const float x[6] = {nx - cx, ny - cy, nz - cz, nx, ny, nz};   // [grouped_xyz - centre ; grouped features = xyz]
      // layer 0 (VALU): features f = 16t + g*4 + r of neighbour c
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
      // layer 1: H1^T (64 x 16) = W1 (64 x 32) H0^T; k index of step s in block t is feature 16t + g*4 + s
      f32x4 h1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4 w = *reinterpret_cast<const float4 *>(sW1 + (16 * u + c) * 32 + 16 * t + g * 4);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, h0[t][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, h0[t][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, h0[t][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, h0[t][3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[u][r] = fmaxf(acc[r] + sb1[16 * u + g * 4 + r], 0.f);
      }
      // layer 2: H2^T (128 x 16) = W2 (128 x 64) H1^T, running max over neighbours of the raw accumulators
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 w = *reinterpret_cast<const float4 *>(sW2 + (16 * v + c) * 64 + 16 * u + g * 4);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, h1[u][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, h1[u][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, h1[u][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, h1[u][3], acc, 0, 0, 0);
        }
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
