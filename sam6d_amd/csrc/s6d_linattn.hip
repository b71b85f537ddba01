// Focused linear attention of the PEM's dense stage, after the projections (gfx950).
//
// Reference: LinearAttention.forward, Pose_Estimation_Model/model/transformer.py:536-564 (kv-first branch: I = 2048 queries against
// J = 196 memory rows, 4 heads of 64):
//     q, k = focus(proj_q(x)), focus(proj_k(mem));  v = proj_v(mem)
//     z  = 1 / (q . sum_j k_j + 1e-6)                                    (B, h, I)
//     kv = k^T v                                                         (B, h, 64, 64)
//     out = merge_heads((q kv) * z)                                      (B, I, 256)
// In library ops that is three head permutes, three batched products, a reduction and five element-wise passes per layer (15
// launches, six layers per PEM pass).  Here: linattn_kv_kernel forms kv and sum_j k_j of every (batch, head) from the focused keys,
// linattn_apply_kernel reads the RAW query projection, applies the focus map (the row norms span the four heads: one LDS exchange
// of partial sums between the block's four waves) and multiplies by kv straight out of SCALAR registers -- a wave owns one head of
// 64 rows, so kv[c][d] is wave-uniform and rides as the scalar operand of v_fmac_f32: no LDS, no staging -- and writes the merged
// (B, I, 256) layout.  fp32 throughout, sequential sums over c and j.
#include "s6d_common.h"

namespace s6d {

constexpr int LA_C = 256, LA_H = 4, LA_D = 64;

#ifdef HIPEMU
#define S6D_LA_CONST(T) T
#else
#define S6D_LA_CONST(T) __attribute__((address_space(4))) T
#endif

// one workgroup per (batch, head): thread t -> kv[c = t >> 2][16 (t & 3) .. + 15];  keys / values staged through LDS 28 rows at a time
__global__ __launch_bounds__(256) void linattn_kv_kernel(const float *__restrict__ kf, long ldk, const float *__restrict__ v, long ldv,
                                                        int J, float *__restrict__ kv, float *__restrict__ ksum) {
  __shared__ float sk[28][LA_D], sv[28][LA_D];
  const int b = blockIdx.x / LA_H, h = blockIdx.x % LA_H, t = threadIdx.x;
  const int c = t >> 2, d0 = (t & 3) * 16;
  float acc[16], ks = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < J; j0 += 28) {
    const int nj = min(28, J - j0);
    __syncthreads();
    for (int e = t; e < nj * LA_D; e += 256) {
      const int jr = e >> 6, cc = e & 63;
      sk[jr][cc] = kf[((size_t)b * J + j0 + jr) * ldk + h * LA_D + cc];
      sv[jr][cc] = v[((size_t)b * J + j0 + jr) * ldv + h * LA_D + cc];
    }
    __syncthreads();
    for (int jr = 0; jr < nj; ++jr) {
      const float kc = sk[jr][c];
      ks += kc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(kc, sv[jr][d0 + i], acc[i]);
    }
  }
  float *o = kv + ((size_t)blockIdx.x * LA_D + c) * LA_D + d0;
#pragma unroll
  for (int i = 0; i < 16; i += 4) *reinterpret_cast<float4 *>(o + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
  if ((t & 3) == 0) ksum[(size_t)blockIdx.x * LA_D + c] = ks;
}

// one workgroup = 64 query rows of one batch element x 4 heads (wave w = head w, lane = row)
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float *__restrict__ xq, const float *__restrict__ inv_scale, int I,
                                                           int power, const float *__restrict__ kv, const float *__restrict__ ksum,
                                                           float *__restrict__ out) {
  __shared__ float part[2][LA_H][64];
  const int nblk = (I + 63) / 64;
  const int b = blockIdx.x / nblk, i0 = (blockIdx.x % nblk) * 64;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i = min(i0 + lane, I - 1);                                  // rows past I are computed, not stored
  const float *src = xq + ((size_t)b * I + i) * LA_C + h * LA_D;
  // focus map (transformer.py:536-547): t = (relu(x) + 1e-6) / softplus(scale); q = t^p * |t| / |t^p| with the norms over all 256 channels
  float q[LA_D];
  float n1 = 0.f, n3 = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < LA_D; c4 += 4) {
    const float4 a = *reinterpret_cast<const float4 *>(src + c4);
    const float4 is = *reinterpret_cast<const float4 *>(inv_scale + h * LA_D + c4);
    const float tv[4] = {(fmaxf(a.x, 0.f) + 1e-6f) * is.x, (fmaxf(a.y, 0.f) + 1e-6f) * is.y, (fmaxf(a.z, 0.f) + 1e-6f) * is.z,
                         (fmaxf(a.w, 0.f) + 1e-6f) * is.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r = tv[e];
      for (int k = 1; k < power; ++k) r *= tv[e];
      n1 = fmaf(tv[e], tv[e], n1);
      n3 = fmaf(r, r, n3);
      q[c4 + e] = r;
    }
  }
  part[0][h][lane] = n1;
  part[1][h][lane] = n3;
  __syncthreads();
  n1 = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
  n3 = (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
  const float f = sqrtf(n1) / sqrtf(n3);
  // (q kv) and q . ksum with kv / ksum of this (batch, head) as scalar operands
  const S6D_LA_CONST(float) *kvs = (const S6D_LA_CONST(float) *)kv + (size_t)(b * LA_H + h) * LA_D * LA_D;
  const S6D_LA_CONST(float) *kss = (const S6D_LA_CONST(float) *)ksum + (size_t)(b * LA_H + h) * LA_D;
  float acc[LA_D], zd = 0.f;
#pragma unroll
  for (int d = 0; d < LA_D; ++d) acc[d] = 0.f;
#pragma unroll
  for (int c = 0; c < LA_D; ++c) {
    const float qc = q[c] * f;
    zd = fmaf(qc, kss[c], zd);
#pragma unroll
    for (int d = 0; d < LA_D; ++d) acc[d] = fmaf(qc, kvs[c * LA_D + d], acc[d]);
  }
  const float z = 1.0f / (zd + 1e-6f);
  if (i0 + lane < I) {
    float *o = out + ((size_t)b * I + i) * LA_C + h * LA_D;
#pragma unroll
    for (int d = 0; d < LA_D; d += 4)
      *reinterpret_cast<float4 *>(o + d) = make_float4(acc[d] * z, acc[d + 1] * z, acc[d + 2] * z, acc[d + 3] * z);
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_linear_attention_f32(const float *q_proj, const float *inv_scale, int power, const float *k_focused, long ldk,
                                        const float *v, long ldv, int B, int I, int J, int C, float *kv_ws, float *out,
                                        void *stream) {
  if (B < 0 || I < 0 || J <= 0 || power < 1 || ldk < C || ldv < C) return S6D_EINVAL;
  if (C != LA_C) return S6D_EUNSUPPORTED;                     // released model: d_model 256 = 4 heads x 64
  if (B == 0 || I == 0) return S6D_OK;
  if (!q_proj || !inv_scale || !k_focused || !v || !kv_ws || !out) return S6D_EINVAL;
  if ((((uintptr_t)q_proj | (uintptr_t)out | (uintptr_t)kv_ws | (uintptr_t)inv_scale) & 15)) return S6D_EINVAL;
  float *kv = kv_ws, *ks = kv_ws + (size_t)B * LA_H * LA_D * LA_D;
  hipLaunchKernelGGL(linattn_kv_kernel, dim3((unsigned)(B * LA_H)), dim3(256), 0, as_stream(stream), k_focused, ldk, v, ldv, J, kv, ks);
  int rc = launch_status();
  if (rc != S6D_OK) return rc;
  const unsigned nblk = (unsigned)((I + 63) / 64);
  hipLaunchKernelGGL(linattn_apply_kernel, dim3((unsigned)B * nblk), dim3(256), 0, as_stream(stream), q_proj, inv_scale, I, power, kv,
                     ks, out);
  return launch_status();
}

extern "C" long s6d_linear_attention_workspace_floats(int B) { return (long)B * LA_H * (LA_D * LA_D + LA_D); }
