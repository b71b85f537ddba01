// Pose-solver kernels of the PEM matching heads (gfx950).
//
// Re-derivation of Pose_Estimation_Model/utils/model_utils.py:
//   compute_coarse_Rt  :187-246  -> pose_hypotheses_kernel (gather 3+3 points, closed-form
//                                   Procrustes, residual) and transform_min_dist_kernel
//   weighted_procrustes :287-363 -> rot_from_h (R = V diag(1,1,det) U^T without a LAPACK SVD)
//   compute_fine_Rt    :250-283  -> transform_min_dist_kernel (P = 1)
// The reference funnels B*6000 3x3 matrices through torch.svd; here every hypothesis is one
// lane doing a Jacobi eigen-solve in registers (fp64: 78 TF/s on MI355X makes it free).
#include "s6d_common.h"
#include "s6d_rot.h"

namespace s6d {

__global__ void rot_from_h_kernel(const float *__restrict__ H, int n, float *__restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double h[9], r[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) h[k] = (double)H[(size_t)i * 9 + k];
  rot_from_h(h, r);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = (float)r[k];
}

// One lane per pose hypothesis (model_utils.py:216-231): pair index -> (i1, i2) = (idx / N2, idx % N2)
// clamped; src = 3 points of pts2, ref = 3 points of pts1, unit weights normalised by (3 + 1e-5);
// R, t by Procrustes; residual = mean_i |(p1_i - t) R - p2_i|.
__global__ void pose_hypotheses_kernel(const float *__restrict__ pts1, const float *__restrict__ pts2,
                                       const int32_t *__restrict__ pair, int N1, int N2, int n_hyp,
                                       float *__restrict__ Rout, float *__restrict__ tout,
                                       float *__restrict__ dis) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (h >= n_hyp) return;
  const float *P1 = pts1 + (size_t)b * N1 * 3;
  const float *P2 = pts2 + (size_t)b * N2 * 3;
  const int32_t *pi = pair + ((size_t)b * n_hyp + h) * 3;
  double p1[3][3], p2[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int id = pi[k];
    int i1 = id / N2, i2 = id % N2;
    i1 = i1 > N1 - 1 ? N1 - 1 : i1;
    i2 = i2 > N2 - 1 ? N2 - 1 : i2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      p1[k][c] = (double)P1[i1 * 3 + c];
      p2[k][c] = (double)P2[i2 * 3 + c];
    }
  }
  const double w = 1.0 / (3.0 + 1e-5);
  double sc[3], rc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    sc[c] = (p2[0][c] + p2[1][c] + p2[2][c]) * w;
    rc[c] = (p1[0][c] + p1[1][c] + p1[2][c]) * w;
  }
  double H[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += (p2[k][a] - sc[a]) * (w * (p1[k][c] - rc[c]));
      H[a * 3 + c] = s;
    }
  double R[9];
  rot_from_h(H, R);
  double t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = rc[r] - (R[r * 3 + 0] * sc[0] + R[r * 3 + 1] * sc[1] + R[r * 3 + 2] * sc[2]);
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double d0 = p1[k][0] - t[0], d1 = p1[k][1] - t[1], d2 = p1[k][2] - t[2];
    double e = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // ((p1 - t) R)[c] = sum_r d_r R[r][c]
      const double x = d0 * R[0 * 3 + c] + d1 * R[1 * 3 + c] + d2 * R[2 * 3 + c] - p2[k][c];
      e += x * x;
    }
    acc += sqrt(e);
  }
  const size_t o = (size_t)b * n_hyp + h;
#pragma unroll
  for (int k = 0; k < 9; ++k) Rout[o * 9 + k] = (float)R[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) tout[o * 3 + k] = (float)t[k];
  dis[o] = (float)(acc / 3.0);
}

// dmin[b,p,n] = min_m | (pts[b,n] - t[b,p]) R[b,p] - model[b,m] |   (model_utils.py:237-239, 273-275)
// The model cloud sits in LDS (broadcast reads); the distance is the direct (x-y)^2 form, which is
// better conditioned than the reference's x^2 - 2xy + y^2 expansion.
// Round 6: the cloud is kept as three coordinate arrays (padded to a multiple of 4 with copies of point 0: a minimum does not see
// them) and four model points are taken per trip -- three 16-byte broadcast reads, the differences / squares / sums as packed fp32
// instructions (two points per issue slot), min3 for the minima: ~4.5 instructions per point.  The loop over an (x, y, z)-interleaved
// array compiled to ~10 per point (a two-wide vectorised body with scalar reads and a remainder test per pair): 300 hypotheses x 196
// points x 1024 model points per instance were 0.47 ms per 32 instances.  Per point the arithmetic is what it was -- ex ex, then
// fma(ey, ey, .), then fma(ez, ez, .) -- so the distances keep their bits.
typedef float md_f32x2 __attribute__((ext_vector_type(2)));
template <int THREADS>
__global__ __launch_bounds__(THREADS) void transform_min_dist_kernel(
    const float *__restrict__ pts, const float *__restrict__ R, const float *__restrict__ t,
    const float *__restrict__ model, int N, int P, int Nm, float *__restrict__ dmin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Np = (Nm + 3) & ~3;
  float *sx = reinterpret_cast<float *>(smem), *sy = sx + Np, *sz = sy + Np;
  const int b = blockIdx.z, p = blockIdx.y;
  const float *mp = model + (size_t)b * Nm * 3;
  for (int i = threadIdx.x; i < Np; i += THREADS) {
    const int m = i < Nm ? i : 0;
    sx[i] = mp[m * 3 + 0];
    sy[i] = mp[m * 3 + 1];
    sz[i] = mp[m * 3 + 2];
  }
  __syncthreads();
  const int n = blockIdx.x * THREADS + threadIdx.x;
  if (n >= N) return;
  const float *Rp = R + ((size_t)b * P + p) * 9;
  const float *tp = t + ((size_t)b * P + p) * 3;
  const float *q = pts + ((size_t)b * N + n) * 3;
  const float d0 = q[0] - tp[0], d1 = q[1] - tp[1], d2 = q[2] - tp[2];
  const float x = d0 * Rp[0] + d1 * Rp[3] + d2 * Rp[6];
  const float y = d0 * Rp[1] + d1 * Rp[4] + d2 * Rp[7];
  const float z = d0 * Rp[2] + d1 * Rp[5] + d2 * Rp[8];
  const md_f32x2 x2 = {x, x}, y2 = {y, y}, z2 = {z, z};
  float best = 3.4e38f;
#pragma unroll 2
  for (int m = 0; m < Np; m += 4) {
    const float4 mx = *reinterpret_cast<const float4 *>(sx + m), my = *reinterpret_cast<const float4 *>(sy + m),
                 mz = *reinterpret_cast<const float4 *>(sz + m);
    const md_f32x2 ex0 = x2 - (md_f32x2){mx.x, mx.y}, ex1 = x2 - (md_f32x2){mx.z, mx.w};
    const md_f32x2 ey0 = y2 - (md_f32x2){my.x, my.y}, ey1 = y2 - (md_f32x2){my.z, my.w};
    const md_f32x2 ez0 = z2 - (md_f32x2){mz.x, mz.y}, ez1 = z2 - (md_f32x2){mz.z, mz.w};
    md_f32x2 e0 = ex0 * ex0, e1 = ex1 * ex1;
    e0 = __builtin_elementwise_fma(ey0, ey0, e0);
    e1 = __builtin_elementwise_fma(ey1, ey1, e1);
    e0 = __builtin_elementwise_fma(ez0, ez0, e0);
    e1 = __builtin_elementwise_fma(ez1, ez1, e1);
    best = fminf(fminf(best, e0.x), e0.y);
    best = fminf(fminf(best, e1.x), e1.y);
  }
  dmin[((size_t)b * P + p) * N + n] = sqrtf(best);
}

// weighted_procrustes for one point set per instance (model_utils.py:287-363 as compute_fine_Rt calls it, :268-271): weights below
// `thresh` are zeroed, w = weights / (sum + eps), sc = sum w src, rc = sum w ref, H = (src - sc)^T (w (ref - rc)), R = rot_from_h(H),
// t = rc - R sc.  ONE workgroup per instance and a FIXED summation order (thread i takes points i, i + 256, ...; a binary tree
// over the 256 partial sums), accumulated in double: the result of an instance does not depend on how many instances share the
// launch -- the library chain it replaces (where, sum, div, 2 x mul + sum, bmm(3 x N, N x 3), matmul) picks reduction / GEMM
// configurations by the BATCH size, so a frame's poses differed in the last bits between a group of frames and a frame alone.
constexpr int kWpThreads = 256;
__device__ __forceinline__ void wp_tree(double *red, int tid, int width) {      // red[width][256] -> sums in red[k * 256]
  for (int s = kWpThreads / 2; s > 0; s >>= 1) {
    __syncthreads();
    if (tid < s)
      for (int k = 0; k < width; ++k) red[k * kWpThreads + tid] += red[k * kWpThreads + tid + s];
  }
  __syncthreads();
}

__global__ __launch_bounds__(kWpThreads) void weighted_procrustes_kernel(const float *__restrict__ src, const float *__restrict__ ref,
                                                                        const float *__restrict__ weights, int N, float thresh,
                                                                        float eps, float *__restrict__ Rout, float *__restrict__ tout) {
  __shared__ double red[9 * kWpThreads];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *S = src + (size_t)b * N * 3, *Q = ref + (size_t)b * N * 3, *Wt = weights + (size_t)b * N;
  auto wt = [&](int i) { const float w = Wt[i]; return w < thresh ? 0.0f : w; };
  double a = 0;
  for (int i = tid; i < N; i += kWpThreads) a += (double)wt(i);
  red[tid] = a;
  wp_tree(red, tid, 1);
  const float wsum = (float)red[0] + eps;                 // the reference's float32 normaliser
  __syncthreads();
  double c[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < N; i += kWpThreads) {
    const double w = (double)(wt(i) / wsum);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      c[k] += w * (double)S[i * 3 + k];
      c[3 + k] += w * (double)Q[i * 3 + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) red[k * kWpThreads + tid] = c[k];
  wp_tree(red, tid, 6);
  float sc[3], rc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    sc[k] = (float)red[k * kWpThreads];
    rc[k] = (float)red[(3 + k) * kWpThreads];
  }
  __syncthreads();
  double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < N; i += kWpThreads) {
    const float w = wt(i) / wsum;
    float ds[3], dq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ds[k] = S[i * 3 + k] - sc[k];
      dq[k] = w * (Q[i * 3 + k] - rc[k]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) h[r * 3 + k] += (double)ds[r] * (double)dq[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) red[k * kWpThreads + tid] = h[k];
  wp_tree(red, tid, 9);
  if (tid == 0) {
    double H[9], R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = (double)(float)red[k * kWpThreads];
    rot_from_h(H, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rout[(size_t)b * 9 + k] = (float)R[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float Rr0 = (float)R[r * 3 + 0], Rr1 = (float)R[r * 3 + 1], Rr2 = (float)R[r * 3 + 2];
      tout[(size_t)b * 3 + r] = rc[r] - (Rr0 * sc[0] + Rr1 * sc[1] + Rr2 * sc[2]);
    }
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_rot_from_h_f32(const float *H, int n, float *R, void *stream) {
  if (n < 0) return S6D_EINVAL;
  if (n == 0) return S6D_OK;
  if (!H || !R) return S6D_EINVAL;
  hipLaunchKernelGGL(rot_from_h_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), H, n, R);
  return launch_status();
}

extern "C" int s6d_pose_hypotheses_f32(const float *pts1, const float *pts2, const int32_t *pair, int B, int N1,
                                       int N2, int n_hyp, float *R, float *t, float *dis, void *stream) {
  if (B < 0 || N1 <= 0 || N2 <= 0 || n_hyp < 0) return S6D_EINVAL;
  if ((size_t)B * n_hyp == 0) return S6D_OK;
  if (!pts1 || !pts2 || !pair || !R || !t || !dis) return S6D_EINVAL;
  hipLaunchKernelGGL(pose_hypotheses_kernel, dim3((n_hyp + 63) / 64, B), dim3(64), 0, as_stream(stream), pts1, pts2,
                     pair, N1, N2, n_hyp, R, t, dis);
  return launch_status();
}

extern "C" int s6d_min_dist_f32(const float *pts, const float *R, const float *t, const float *model, int B, int N,
                                int P, int Nm, float *dmin, void *stream) {
  if (B < 0 || N <= 0 || P < 0 || Nm <= 0) return S6D_EINVAL;
  if ((size_t)B * P == 0) return S6D_OK;
  if (!pts || !R || !t || !model || !dmin) return S6D_EINVAL;
  const size_t lds = (size_t)((Nm + 3) & ~3) * 12;
  if (lds > 64 * 1024) return S6D_EUNSUPPORTED;
  dim3 grid((N + 255) / 256, P, B);
  hipLaunchKernelGGL((transform_min_dist_kernel<256>), grid, dim3(256), lds, as_stream(stream), pts, R, t, model, N, P, Nm, dmin);
  return launch_status();
}

extern "C" int s6d_weighted_procrustes_f32(const float *src, const float *ref, const float *weights, int B, int N, float weight_thresh,
                                           float eps, float *R, float *t, void *stream) {
  if (B < 0 || N <= 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!src || !ref || !weights || !R || !t) return S6D_EINVAL;
  hipLaunchKernelGGL(weighted_procrustes_kernel, dim3(B), dim3(kWpThreads), 0, as_stream(stream), src, ref, weights, N, weight_thresh,
                     eps, R, t);
  return launch_status();
}
