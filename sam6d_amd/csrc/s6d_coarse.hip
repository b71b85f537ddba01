// Coarse pose hypotheses of the PEM (gfx950): the sampling head and the two selections of compute_coarse_Rt as three small
// workgroup-per-instance kernels (the reference and round 1 of this repo run them as ~15 library launches, a sort included).
//
// Reference: Pose_Estimation_Model/utils/model_utils.py
//   :203-212  score = softmax(atten, 2) * softmax(atten, 1); labels by arg-max against the background row / column; mask
//   :213-219  score[:, 1:, 1:] ** 1.5 -> cumsum over the flattened 196 x 196 bins -> / (last + 1e-8) -> searchsorted of 3 x 6000
//             uniforms (drawn by the caller here: BASELINE's parity needs the random numbers to be an input)
//   :233      topk(dis, 300, largest=False) of the 6000 hypothesis residuals, gathers of R, t
//   :240-246  score = sum(w1) / (sum(dmin * w1) + 1e-8) per hypothesis, arg-max, gathers
// (between them: s6d_pose_hypotheses_f32 and s6d_min_dist_f32, csrc/s6d_pose.hip.)
//
// coarse_sample_kernel, one 1024-thread workgroup per instance:
//   pass 1  e = exp(a) (|a| <= 1/temp = 10: no max shift), row sums by one wave per row, column sums as per-wave partials folded
//           in wave order (deterministic)
//   pass 2  p = (e / r_i)(e / c_j); row arg-max -> w1, column arg-max -> w2 (first index wins, as torch.max)
//   pass 3  q = (p w1 w2)^1.5 for i, j >= 1 into LDS (196 x 196 floats = 150 KiB: the whole distribution stays on chip)
//   scan    inclusive prefix sum over q in row-major order, accumulated in float64 (per-thread run of 38 bins, wave scan by
//           shuffles, 16 wave totals through LDS), stored back as float32 and normalised by (total + 1e-8) like the reference
//   search  every uniform by binary search (lower bound = torch.searchsorted's default side) over the LDS array
// Parity: the bins are float32 like the reference's; the prefix sums differ from a sequential float32 cumsum (ATen's CPU kernel
// accumulates in float64 too) in the last bit at most, so a uniform picks a different bin only when it falls within one ulp of a
// bin boundary; the tests compare the sampled pairs and the final pose.
#include "s6d_common.h"

namespace s6d {

constexpr int CS_THREADS = 1024;
constexpr int CS_WAVES = CS_THREADS / 64;

extern __shared__ __attribute__((aligned(16))) char cs_smem[];

__global__ __launch_bounds__(CS_THREADS) void coarse_sample_kernel(const float *__restrict__ atten, const float *__restrict__ rand_u,
                                                                  int M1, int M2, int n_u, int32_t *__restrict__ pair,
                                                                  float *__restrict__ w1_out) {
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int L = (M1 - 1) * (M2 - 1);
  float *q = reinterpret_cast<float *>(cs_smem);                        // [L] bins, later their normalised prefix sums
  // the per-wave scratch of passes 1 / 2 aliases the bins; with few rows (M1 - 1 < 32) it is the larger of the two, so the
  // small arrays start behind whichever is longer (they used to start at q + L and were overrun for N1 <= 32)
  const int Lq = max(L, CS_WAVES * M2 * 2);
  float *rs = q + Lq;                                                    // [M1] row sums
  float *cs = rs + M1;                                                   // [M2] column sums
  float *w1 = cs + M2;                                                   // [M1] (entry 0 unused)
  float *w2 = w1 + M1;                                                   // [M2]
  double *wt = reinterpret_cast<double *>(cs_smem + (((size_t)(Lq + 2 * M1 + 2 * M2) * 4 + 7) & ~(size_t)7));   // [CS_WAVES]
  // scratch of passes 1 / 2 (per-wave column partials) aliases the bin array, which is only written in pass 3
  float *part = q;                                                       // [CS_WAVES][M2]
  unsigned long long *kpart = reinterpret_cast<unsigned long long *>(q);  // [CS_WAVES][M2]
  const float *A = atten + (size_t)b * M1 * M2;
  const int ncol = (M2 + 63) / 64;                                       // columns per lane (<= 4 checked by the launcher)

  // ---- pass 1: sums -------------------------------------------------------------------------------------------------------------
  float cacc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < M1; i += CS_WAVES) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      if (k < ncol && j < M2) {
        const float e = __expf(A[(size_t)i * M2 + j]);
        r += e;
        cacc[k] += e;
      }
    }
    r = wave_sum(r);
    if (lane == 0) rs[i] = r;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane + 64 * k;
    if (k < ncol && j < M2) part[wave * M2 + j] = cacc[k];
  }
  __syncthreads();
  for (int j = tid; j < M2; j += CS_THREADS) {
    float s = 0.f;
    for (int w = 0; w < CS_WAVES; ++w) s += part[w * M2 + j];            // fixed order
    cs[j] = s;
  }
  __syncthreads();

  // ---- pass 2: labels -----------------------------------------------------------------------------------------------------------
  unsigned long long ck[4] = {0ull, 0ull, 0ull, 0ull};
  float ic[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane + 64 * k;
    ic[k] = (k < ncol && j < M2) ? 1.0f / cs[j] : 0.f;
  }
  for (int i = wave; i < M1; i += CS_WAVES) {
    const float ir = 1.0f / rs[i];
    unsigned long long rk = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      if (k < ncol && j < M2) {
        const float e = __expf(A[(size_t)i * M2 + j]);
        const float p = (e * ir) * (e * ic[k]);
        // p >= 0: the bit pattern orders like the value; the complemented index makes the FIRST index win ties
        const unsigned long long kr = ((unsigned long long)__float_as_uint(p) << 32) | (0xffffffffu - (unsigned)j);
        const unsigned long long kc = ((unsigned long long)__float_as_uint(p) << 32) | (0xffffffffu - (unsigned)i);
        rk = kr > rk ? kr : rk;
        ck[k] = kc > ck[k] ? kc : ck[k];
      }
    }
    rk = wave_max_u64(rk);
    if (lane == 0) w1[i] = (0xffffffffu - (unsigned)(rk & 0xffffffffull)) > 0u ? 1.f : 0.f;
  }
  __syncthreads();                                                       // `part` is dead, `kpart` takes its place
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane + 64 * k;
    if (k < ncol && j < M2) kpart[wave * M2 + j] = ck[k];
  }
  __syncthreads();
  for (int j = tid; j < M2; j += CS_THREADS) {
    unsigned long long m = 0ull;
    for (int w = 0; w < CS_WAVES; ++w) {
      const unsigned long long v = kpart[w * M2 + j];
      m = v > m ? v : m;
    }
    w2[j] = (0xffffffffu - (unsigned)(m & 0xffffffffull)) > 0u ? 1.f : 0.f;
  }
  __syncthreads();
  for (int i = 1 + tid; i < M1; i += CS_THREADS) w1_out[(size_t)b * (M1 - 1) + (i - 1)] = w1[i];

  // ---- pass 3: bins -------------------------------------------------------------------------------------------------------------
  for (int i = 1 + wave; i < M1; i += CS_WAVES) {
    const float ir = 1.0f / rs[i], wi = w1[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      if (k < ncol && j >= 1 && j < M2) {
        const float e = __expf(A[(size_t)i * M2 + j]);
        const float s = ((e * ir) * (e * ic[k])) * wi * w2[j];
        q[(i - 1) * (M2 - 1) + (j - 1)] = s * sqrtf(s);                 // s ** 1.5
      }
    }
  }
  __syncthreads();

  // ---- inclusive scan in float64 ---------------------------------------------------------------------------------------------------
  const int per = (L + CS_THREADS - 1) / CS_THREADS;
  const int i0 = min(tid * per, L), i1 = min(i0 + per, L);
  double run = 0.0;
  for (int i = i0; i < i1; ++i) run += (double)q[i];
  double incl = run;                                                     // inclusive scan of the thread totals inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl(incl, lane >= d ? lane - d : 0);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wt[wave] = incl;
  __syncthreads();
  double base = incl - run;
  for (int w = 0; w < wave; ++w) base += wt[w];                          // fixed order
  double total = 0.0;
  for (int w = 0; w < CS_WAVES; ++w) total += wt[w];
  const float denom = (float)total + 1e-8f;                              // cum[:, -1] + 1e-8 in float32
  {
    double acc = base;
    for (int i = i0; i < i1; ++i) {
      acc += (double)q[i];
      q[i] = (float)acc / denom;
    }
  }
  __syncthreads();

  // ---- searchsorted (left): first index with q[idx] >= u ------------------------------------------------------------------------
  const float *U = rand_u + (size_t)b * n_u;
  for (int t = tid; t < n_u; t += CS_THREADS) {
    const float u = U[t];
    int lo = 0, hi = L;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (q[mid] < u) lo = mid + 1; else hi = mid;
    }
    pair[(size_t)b * n_u + t] = lo;
  }
}

// The k smallest of n residuals per instance in ascending (value, index) order, with the rows of R and t they select
// (topk(largest=False) + two gathers, model_utils.py:233-235).  One workgroup per instance: 64-bit keys (value bits, index),
// bitonic sort of the padded power of two in LDS.
__global__ __launch_bounds__(1024) void smallest_k_kernel(const float *__restrict__ dis, const float *__restrict__ Rs,
                                                         const float *__restrict__ ts, int n, int npad, int k,
                                                         float *__restrict__ Rk, float *__restrict__ tk, int32_t *__restrict__ idx) {
  unsigned long long *key = reinterpret_cast<unsigned long long *>(cs_smem);
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < npad; i += 1024) {
    unsigned long long v = ~0ull;                                        // padding sorts last
    if (i < n) {
      const float d = dis[(size_t)b * n + i];
      // residuals are >= 0 (a mean of norms); a NaN would sort behind every number, like torch.topk(largest=False)
      v = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
    }
    key[i] = v;
  }
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (npad >> 1); t += 1024) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = key[lo], c = key[hi];
        if ((a > c) == up) {
          key[lo] = c;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int j = tid; j < k; j += 1024) {
    const int src = (int)(key[j] & 0xffffffffull);
    idx[(size_t)b * k + j] = src;
    const float *r = Rs + ((size_t)b * n + src) * 9;
    float *ro = Rk + ((size_t)b * k + j) * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) ro[e] = r[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) tk[((size_t)b * k + j) * 3 + e] = ts[((size_t)b * n + src) * 3 + e];
  }
}

// score_p = sum(w1) / (sum_n dmin[p,n] w1[n] + 1e-8); the first arg-max; its R, t  (model_utils.py:240-246).
// One workgroup per instance, one wave per hypothesis at a time.
__global__ __launch_bounds__(256) void hypothesis_select_kernel(const float *__restrict__ dmin, const float *__restrict__ w1,
                                                               const float *__restrict__ Rk, const float *__restrict__ tk, int P,
                                                               int N, float *__restrict__ R, float *__restrict__ t) {
  __shared__ unsigned long long best[4];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *W = w1 + (size_t)b * N;
  float wsum = 0.f;
  for (int n = lane; n < N; n += 64) wsum += W[n];
  wsum = wave_sum(wsum);
  unsigned long long bk = 0ull;
  for (int p = wave; p < P; p += 4) {
    const float *d = dmin + ((size_t)b * P + p) * N;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += d[n] * W[n];
    s = wave_sum(s);
    const float sc = wsum / (s + 1e-8f);                                 // >= 0
    const unsigned long long kk = ((unsigned long long)__float_as_uint(sc) << 32) | (0xffffffffu - (unsigned)p);
    bk = kk > bk ? kk : bk;
  }
  if (lane == 0) best[wave] = bk;
  __syncthreads();
  if (threadIdx.x < 12) {
    unsigned long long m = best[0];
    for (int w = 1; w < 4; ++w) m = best[w] > m ? best[w] : m;
    const int p = (int)(0xffffffffu - (unsigned)(m & 0xffffffffull));
    const int e = threadIdx.x;
    if (e < 9) R[(size_t)b * 9 + e] = Rk[((size_t)b * P + p) * 9 + e];
    else t[(size_t)b * 3 + (e - 9)] = tk[((size_t)b * P + p) * 3 + (e - 9)];
  }
}

}  // namespace s6d

using namespace s6d;

static size_t cs_lds_bytes(int M1, int M2) {
  const size_t L = (size_t)(M1 - 1) * (M2 - 1);
  const size_t scratch = (size_t)CS_WAVES * M2 * 2;                      // floats: the u64 column partials alias the bins
  size_t bins = ((L > scratch ? L : scratch) + 2 * (size_t)M1 + 2 * (size_t)M2) * 4;   // same layout as the kernel's
  bins = (bins + 7) & ~(size_t)7;
  return bins + CS_WAVES * 8;
}

extern "C" int s6d_coarse_sample_f32(const float *atten, const float *rand_u, int B, int M1, int M2, int n_u, int32_t *pair,
                                     float *w1, void *stream) {
  if (B < 0 || M1 < 2 || M2 < 2 || n_u < 0) return S6D_EINVAL;
  if (M2 > 256 || cs_lds_bytes(M1, M2) > 160 * 1024) return S6D_EUNSUPPORTED;
  if (B == 0) return S6D_OK;
  if (!atten || !rand_u || !pair || !w1) return S6D_EINVAL;
  const size_t lds = cs_lds_bytes(M1, M2);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&coarse_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL(coarse_sample_kernel, dim3(B), dim3(CS_THREADS), lds, as_stream(stream), atten, rand_u, M1, M2, n_u, pair, w1);
  return launch_status();
}

extern "C" int s6d_smallest_k_f32(const float *dis, const float *Rs, const float *ts, int B, int n, int k, float *Rk, float *tk,
                                  int32_t *idx, void *stream) {
  if (B < 0 || n <= 0 || k <= 0 || k > n) return S6D_EINVAL;
  int npad = 2;
  while (npad < n) npad <<= 1;
  if ((size_t)npad * 8 > 128 * 1024) return S6D_EUNSUPPORTED;
  if (B == 0) return S6D_OK;
  if (!dis || !Rs || !ts || !Rk || !tk || !idx) return S6D_EINVAL;
  const size_t lds = (size_t)npad * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&smallest_k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(smallest_k_kernel, dim3(B), dim3(1024), lds, as_stream(stream), dis, Rs, ts, n, npad, k, Rk, tk, idx);
  return launch_status();
}

extern "C" int s6d_hypothesis_select_f32(const float *dmin, const float *w1, const float *Rk, const float *tk, int B, int P, int N,
                                         float *R, float *t, void *stream) {
  if (B < 0 || P <= 0 || N <= 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!dmin || !w1 || !Rk || !tk || !R || !t) return S6D_EINVAL;
  hipLaunchKernelGGL(hypothesis_select_kernel, dim3(B), dim3(256), 0, as_stream(stream), dmin, w1, Rk, tk, P, N, R, t);
  return launch_status();
}
