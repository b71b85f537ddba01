// Soft-assignment head of the fine point matching (gfx950): dual softmax, background masking, row-normalised
// assignment applied to the template cloud -- compute_fine_Rt, Pose_Estimation_Model/utils/model_utils.py:262-270.
//
// Reference dataflow on the (B, 2049, 2049) fp32 similarity (16.8 MB per instance): softmax(dim=2), softmax(dim=1),
// product, two arg-maxes, two mask multiplies, row sum, divide, (2048x2048)@(2048x3): >= 8 full passes.
// Here: three streaming passes, each recomputing e = exp(s) in registers (|s| <= 1/temp = 10, so no max shift is
// needed: sums stay below 2049*e^10 ~ 4.5e7):
//   pass A  row sums r_i = sum_j e_ij; per-tile column partial sums -> c_j (deterministic two-stage reduction)
//   pass B  p_ij = e_ij^2 / (r_i c_j); row arg-max (-> w1_i = label1 > 0); per-tile column arg-max partials (-> w2_j)
//   pass C  a_ij = p_ij w1_i w2_j (i,j >= 1);  wsum_i = sum_j a_ij;  pred_i = sum_j a_ij pts2_j / (wsum_i + 1e-6)
// One wavefront owns whole rows (lane <-> columns lane+64k, coalesced), so every row reduction is shuffle-only;
// column quantities accumulate in registers across the wave's rows and meet the other waves' in LDS once per tile.
#include "s6d_common.h"

namespace s6d {

constexpr int AS_KC = 33;                 // columns per lane: supports M2 <= 64*33 = 2112 (fine matching: 2049)
constexpr int AS_ROWS_PER_WAVE = 4;
constexpr int AS_WAVES = 4;
constexpr int AS_TILE = AS_ROWS_PER_WAVE * AS_WAVES;   // rows per workgroup

__device__ __forceinline__ unsigned long long argmax_key(float v, int idx) {   // v >= 0; first index wins ties
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned)idx);
}

// ---- pass A -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AS_WAVES * 64) void assign_sums_kernel(const float *__restrict__ S, int M1, int M2,
                                                                   float *__restrict__ rsum, float *__restrict__ cpart) {
  __shared__ float sc[AS_WAVES][64 * AS_KC];
  const int b = blockIdx.y, tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *Sb = S + (size_t)b * M1 * M2;
  float cs[AS_KC];
#pragma unroll
  for (int k = 0; k < AS_KC; ++k) cs[k] = 0.f;
  for (int rr = 0; rr < AS_ROWS_PER_WAVE; ++rr) {
    const int i = tile * AS_TILE + wave * AS_ROWS_PER_WAVE + rr;
    if (i >= M1) break;                                             // wave-uniform
    const float *row = Sb + (size_t)i * M2;
    float rs = 0.f;
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) {
      const int j = lane + 64 * k;
      const float e = j < M2 ? __expf(row[j]) : 0.f;
      rs += e;
      cs[k] += e;
    }
    rs = wave_sum(rs);
    if (lane == 0) rsum[(size_t)b * M1 + i] = rs;
  }
#pragma unroll
  for (int k = 0; k < AS_KC; ++k) sc[wave][lane + 64 * k] = cs[k];
  __syncthreads();
  const int ntile = gridDim.x;
  for (int j = threadIdx.x; j < M2; j += AS_WAVES * 64)
    cpart[((size_t)b * ntile + tile) * M2 + j] = (sc[0][j] + sc[1][j]) + (sc[2][j] + sc[3][j]);
}

__global__ void assign_colsum_kernel(const float *__restrict__ cpart, int ntile, int M2, float *__restrict__ csum) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M2) return;
  float s = 0.f;
  for (int t = 0; t < ntile; ++t) s += cpart[((size_t)b * ntile + t) * M2 + j];     // fixed order
  csum[(size_t)b * M2 + j] = s;
}

// ---- pass B -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AS_WAVES * 64) void assign_labels_kernel(const float *__restrict__ S, int M1, int M2,
                                                                     const float *__restrict__ rsum,
                                                                     const float *__restrict__ csum,
                                                                     float *__restrict__ w1,
                                                                     unsigned long long *__restrict__ kpart) {
  __shared__ unsigned long long sk[2][64 * AS_KC];                  // 33 KB: waves are folded pairwise
  const int b = blockIdx.y, tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *Sb = S + (size_t)b * M1 * M2;
  float ic[AS_KC];
  unsigned long long ck[AS_KC];
#pragma unroll
  for (int k = 0; k < AS_KC; ++k) {
    const int j = lane + 64 * k;
    ic[k] = j < M2 ? 1.0f / csum[(size_t)b * M2 + j] : 0.f;
    ck[k] = 0ull;
  }
  for (int rr = 0; rr < AS_ROWS_PER_WAVE; ++rr) {
    const int i = tile * AS_TILE + wave * AS_ROWS_PER_WAVE + rr;
    if (i >= M1) break;
    const float *row = Sb + (size_t)i * M2;
    const float ir = 1.0f / rsum[(size_t)b * M1 + i];
    unsigned long long rk = 0ull;
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) {
      const int j = lane + 64 * k;
      if (j < M2) {
        const float e = __expf(row[j]);
        const float p = (e * ir) * (e * ic[k]);                     // softmax(dim=2) * softmax(dim=1)
        const unsigned long long kr = argmax_key(p, j), kc = argmax_key(p, i);
        rk = kr > rk ? kr : rk;
        ck[k] = kc > ck[k] ? kc : ck[k];
      }
    }
    rk = wave_max_u64(rk);
    if (lane == 0 && i >= 1) {
      const int arg = (int)(0xffffffffu - (unsigned)(rk & 0xffffffffull));
      w1[(size_t)b * (M1 - 1) + (i - 1)] = arg > 0 ? 1.f : 0.f;     // label1 > 0 (model_utils.py:264,267)
    }
  }
  // fold waves 2,3 into 0,1, then wave 1 into 0 (max is order independent: deterministic)
  if (wave >= 2) {
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) sk[wave - 2][lane + 64 * k] = ck[k];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) {
      const unsigned long long o = sk[wave][lane + 64 * k];
      ck[k] = o > ck[k] ? o : ck[k];
    }
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) sk[0][lane + 64 * k] = ck[k];
  }
  __syncthreads();
  if (wave == 0) {
    const int ntile = gridDim.x;
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) {
      const int j = lane + 64 * k;
      const unsigned long long o = sk[0][j];
      if (j < M2) kpart[((size_t)b * ntile + tile) * M2 + j] = o > ck[k] ? o : ck[k];
    }
  }
}

__global__ void assign_collabel_kernel(const unsigned long long *__restrict__ kpart, int ntile, int M2,
                                       float *__restrict__ w2) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M2) return;
  unsigned long long m = 0ull;
  for (int t = 0; t < ntile; ++t) {
    const unsigned long long v = kpart[((size_t)b * ntile + t) * M2 + j];
    m = v > m ? v : m;
  }
  const int arg = (int)(0xffffffffu - (unsigned)(m & 0xffffffffull));
  w2[(size_t)b * M2 + j] = (j >= 1 && arg > 0) ? 1.f : 0.f;        // label2 > 0 (model_utils.py:265,267); col 0 unused
}

// ---- pass C -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AS_WAVES * 64) void assign_apply_kernel(const float *__restrict__ S, int M1, int M2,
                                                                    const float *__restrict__ rsum,
                                                                    const float *__restrict__ csum,
                                                                    const float *__restrict__ w1,
                                                                    const float *__restrict__ w2,
                                                                    const float *__restrict__ pts2,
                                                                    float *__restrict__ pred, float *__restrict__ wsum) {
  const int b = blockIdx.y, tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *Sb = S + (size_t)b * M1 * M2;
  const float *P2 = pts2 + (size_t)b * (M2 - 1) * 3;
  float f[AS_KC];                                                   // w2_j / c_j (0 for j = 0 and j >= M2)
#pragma unroll
  for (int k = 0; k < AS_KC; ++k) {
    const int j = lane + 64 * k;
    f[k] = (j >= 1 && j < M2) ? w2[(size_t)b * M2 + j] / csum[(size_t)b * M2 + j] : 0.f;
  }
  for (int rr = 0; rr < AS_ROWS_PER_WAVE; ++rr) {
    const int i = tile * AS_TILE + wave * AS_ROWS_PER_WAVE + rr;
    if (i >= M1) break;
    if (i == 0) continue;
    const size_t o = (size_t)b * (M1 - 1) + (i - 1);
    if (w1[o] == 0.f) {                                             // whole row masked (wave-uniform)
      if (lane == 0) { wsum[o] = 0.f; pred[o * 3] = 0.f; pred[o * 3 + 1] = 0.f; pred[o * 3 + 2] = 0.f; }
      continue;
    }
    const float *row = Sb + (size_t)i * M2;
    const float ir = 1.0f / rsum[(size_t)b * M1 + i];
    float ws = 0.f, px = 0.f, py = 0.f, pz = 0.f;
#pragma unroll
    for (int k = 0; k < AS_KC; ++k) {
      const int j = lane + 64 * k;
      if (j >= 1 && j < M2) {
        const float e = __expf(row[j]);
        const float a = (e * ir) * (e * f[k]);
        const float *q = P2 + (size_t)(j - 1) * 3;
        ws += a;
        px += a * q[0];
        py += a * q[1];
        pz += a * q[2];
      }
    }
    ws = wave_sum(ws); px = wave_sum(px); py = wave_sum(py); pz = wave_sum(pz);
    if (lane == 0) {
      const float inv = 1.0f / (ws + 1e-6f);                        // model_utils.py:268
      wsum[o] = ws;
      pred[o * 3] = px * inv;
      pred[o * 3 + 1] = py * inv;
      pred[o * 3 + 2] = pz * inv;
    }
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" long s6d_fine_assign_workspace_bytes(int B, int M1, int M2) {
  const long ntile = (M1 + AS_TILE - 1) / AS_TILE;
  // rsum | csum | w2 | cpart (floats)  +  kpart (u64)
  return ((long)B * M1 + 2L * B * M2 + (long)B * ntile * M2) * 4 + (long)B * ntile * M2 * 8 + 64;
}

extern "C" int s6d_fine_assign_f32(const float *atten, const float *pts2, int B, int M1, int M2, void *workspace,
                                   float *pred, float *wsum, float *w1, void *stream) {
  if (B < 0 || M1 < 2 || M2 < 2) return S6D_EINVAL;
  if (M2 > 64 * AS_KC) return S6D_EUNSUPPORTED;
  if (B == 0) return S6D_OK;
  if (!atten || !pts2 || !workspace || !pred || !wsum || !w1) return S6D_EINVAL;
  const int ntile = (M1 + AS_TILE - 1) / AS_TILE;
  float *rsum = reinterpret_cast<float *>(workspace);
  float *csum = rsum + (size_t)B * M1;
  float *w2 = csum + (size_t)B * M2;
  float *cpart = w2 + (size_t)B * M2;
  uintptr_t kp = reinterpret_cast<uintptr_t>(cpart + (size_t)B * ntile * M2);
  kp = (kp + 7) & ~(uintptr_t)7;
  unsigned long long *kpart = reinterpret_cast<unsigned long long *>(kp);
  hipStream_t st = as_stream(stream);
  dim3 grid(ntile, B), colgrid((M2 + 255) / 256, B);
  hipLaunchKernelGGL(assign_sums_kernel, grid, dim3(AS_WAVES * 64), 0, st, atten, M1, M2, rsum, cpart);
  hipLaunchKernelGGL(assign_colsum_kernel, colgrid, dim3(256), 0, st, cpart, ntile, M2, csum);
  hipLaunchKernelGGL(assign_labels_kernel, grid, dim3(AS_WAVES * 64), 0, st, atten, M1, M2, rsum, csum, w1, kpart);
  hipLaunchKernelGGL(assign_collabel_kernel, colgrid, dim3(256), 0, st, kpart, ntile, M2, w2);
  hipLaunchKernelGGL(assign_apply_kernel, grid, dim3(AS_WAVES * 64), 0, st, atten, M1, M2, rsum, csum, w1, w2, pts2, pred,
                     wsum);
  return launch_status();
}
