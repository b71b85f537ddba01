// Kernels of the PEM per-detection pre-processing (host side: sam6d_amd/pem/preprocess.py).
//
//   sample_indices   the point sampler of get_test_data / get_instance (Pose_Estimation_Model/run_inference_custom.py:224-229,
//                    provider/bop_test_dataset.py:140-145) in its DEFINED form (preprocess.py header: injected uniforms
//                    instead of numpy's global RNG): per detection with n candidate points and keys u_0 .. u_{n-1},
//                      n <= n_sample : idx_i = floor(u_i * n) for i < n_sample            (with replacement)
//                      n >  n_sample : the positions of the n_sample smallest (u, position) pairs, ascending
//                                      (without replacement; == the first n_sample entries of a stable argsort).
//                    The library version is one top-k over a (P, L) table of 64-bit composite keys -- a full sort of up to
//                    3e5 entries per detection.  Here one workgroup per detection reads its keys twice: a 4096-bin histogram of
//                    the keys' leading bits finds the bin in which the n_sample-th smallest key lies, the (at most 4096) keys
//                    up to that bin are collected in LDS and sorted there (bitonic, 64-bit composite key << 32 | position).
//                    Exact, deterministic, independent of the order of the atomics.  Keys must be non-negative floats (their
//                    bit patterns then order like their values); if more than 4096 keys share the leading bits up to the
//                    threshold bin (heavily duplicated keys, never uniform ones) the detection is flagged in `overflow` and the
//                    caller uses the library path for it.
#include "s6d_common.h"

namespace s6d {

constexpr int kSelThreads = 1024;
constexpr int kSelBins = 4096;        // leading 12 bits of the float (sign is 0: 8 exponent + 3 mantissa bits ... see below)
constexpr int kSelCap = 4096;         // candidates sorted in LDS (32 KB of 64-bit composites)

// bin of a non-negative float: its 12 leading bits below the sign (exponent + 4 mantissa bits)
__device__ __forceinline__ unsigned sel_bin(unsigned bits) {
  const unsigned b = bits >> 19;
  return b < (unsigned)kSelBins ? b : (unsigned)kSelBins - 1;    // negative / NaN keys are outside the contract; no wild index
}

__global__ __launch_bounds__(kSelThreads) void sample_indices_kernel(const float *__restrict__ keys, long key_stride,
                                                                    const long *__restrict__ count, int n_sample,
                                                                    long *__restrict__ idx, int *__restrict__ overflow) {
  __shared__ unsigned hist[kSelBins];
  __shared__ unsigned long long cand[kSelCap];
  __shared__ unsigned s_thr, s_ncand;
  const int p = blockIdx.x, tid = threadIdx.x;
  const long n = count[p];
  const float *k = keys + (size_t)p * key_stride;
  long *out = idx + (size_t)p * n_sample;
  if (n <= n_sample) {                                   // with replacement (also n == 0: every index 0)
    for (int i = tid; i < n_sample; i += kSelThreads) out[i] = (long)floor((double)k[i] * (double)n);
    if (tid == 0) overflow[p] = 0;
    return;
  }
  for (int i = tid; i < kSelBins; i += kSelThreads) hist[i] = 0;
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  for (long i = tid; i < n; i += kSelThreads) atomicAdd(&hist[sel_bin(__float_as_uint(k[i]))], 1u);
  __syncthreads();
  if (tid == 0) {                                        // the bin holding the n_sample-th smallest key
    unsigned cum = 0, t = 0;
    for (; t < kSelBins; ++t) {
      cum += hist[t];
      if (cum >= (unsigned)n_sample) break;
    }
    s_thr = t;
    if (cum > (unsigned)kSelCap) s_ncand = 0xffffffffu;  // too many keys up to the threshold bin for the LDS sort
  }
  __syncthreads();
  const unsigned thr = s_thr;
  if (s_ncand == 0xffffffffu) {
    if (tid == 0) overflow[p] = 1;
    return;
  }
  for (int i = tid; i < kSelCap; i += kSelThreads) cand[i] = ~0ull;
  __syncthreads();
  for (long i = tid; i < n; i += kSelThreads) {
    const unsigned b = __float_as_uint(k[i]);
    if (sel_bin(b) <= thr) cand[atomicAdd(&s_ncand, 1u)] = ((unsigned long long)b << 32) | (unsigned long long)i;
  }
  __syncthreads();
  // bitonic sort of kSelCap 64-bit composites, ascending (the padding ~0 sorts last)
  for (int size = 2; size <= kSelCap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < kSelCap / 2; t += kSelThreads) {
        const int lo = 2 * t - (t & (stride - 1));        // index of the lower element of pair t at this stride
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = cand[lo], c = cand[hi];
        if ((a > c) == up) {
          cand[lo] = c;
          cand[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n_sample; i += kSelThreads) out[i] = (long)(cand[i] & 0xffffffffull);
  if (tid == 0) overflow[p] = 0;
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_pem_sample_indices_f32(const float *keys, long key_stride, const int64_t *count, int P, int n_sample,
                                          int64_t *idx, int32_t *overflow, void *stream) {
  if (P < 0 || n_sample <= 0 || n_sample > kSelCap / 2 || key_stride < n_sample) return S6D_EINVAL;
  if (P == 0) return S6D_OK;
  if (!keys || !count || !idx || !overflow) return S6D_EINVAL;
  hipLaunchKernelGGL(sample_indices_kernel, dim3((unsigned)P), dim3(kSelThreads), 0, as_stream(stream), keys, key_stride,
                     (const long *)count, n_sample, (long *)idx, overflow);
  return launch_status();
}
