// Shared host/device helpers for libsam6d_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sam6d_hip.h"

namespace s6d {

constexpr int kWave = 64;

void set_hip_error(hipError_t e);

// Checks the launch that was just enqueued; reports instead of exiting.
inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_hip_error(e);
    return S6D_ELAUNCH;
  }
  return S6D_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
extern int g_s6d_persistent_grid_limit;      // csrc/s6d_capi.hip: s6d_set_persistent_grid_limit

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), o);
    unsigned hi = __shfl_xor((unsigned)(v >> 32), o);
    unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w > v ? w : v;
  }
  return v;
}

// erf-form GELU (nn.GELU(), approximate='none'):  x Phi(x) = relu(x) - |x| Q(|x|),  Q(z) = erfc(z / sqrt 2) / 2 = 2^P(z) with P the
// degree-7 minimax fit of log2 Q on [0, 6] (tools/probes/gelu_fit.py; Q(6) = 1e-9: beyond it the tail is clamped, |error| < 3e-7 up
// to |x| = 300).  Relative error of the result <= 6e-6 on both signs, 600 times inside the bf16 rounding of the output -- and smaller
// than round 2's form (Abramowitz-Stegun 7.1.26, absolute 1.5e-7 on erfc, i.e. up to 1.6e-3 RELATIVE where x < 0 and the result is
// small): on random fp32 arguments 0.16 % of the bf16 results differ from the correctly rounded ones (was 0.75 %).  9 plain VALU +
// one v_exp per element instead of 12 + v_exp + v_rcp.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float z = fminf(ax, 6.0f);
  float p = fmaf(z, -1.833880219e-06f, 6.155846495e-05f);
  p = fmaf(z, p, -9.300006204e-04f);
  p = fmaf(z, p, 8.504784666e-03f);
  p = fmaf(z, p, -5.395101011e-02f);
  p = fmaf(z, p, -4.584769309e-01f);
  p = fmaf(z, p, -1.151244164e+00f);
  p = fmaf(z, p, -9.999961853e-01f);
  return fmaf(-ax, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
// gelu_erf of several values at once, bit for bit (round 5; gelu_erf above is the definition, the kernels call gelu_erf4): the seven fused multiply-adds of the polynomial as packed fp32 instructions
// (v_pk_fma_f32: two IEEE fmas per lane and issue slot, constants broadcast from one scalar register) -- 6.5 plain slots + v_exp per
// element instead of 10 + v_exp.  |x| stays a source modifier of the scalar instructions (the packed ones have none: a packed
// final product would pay a v_and per element).  The lin1 + GELU epilogue is 128 evaluations per lane and tile with no matrix work
// beside it.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ V gelu_erf_poly(V z) {
  V p = __builtin_elementwise_fma(z, (V)(-1.833880219e-06f), (V)(6.155846495e-05f));
  p = __builtin_elementwise_fma(z, p, (V)(-9.300006204e-04f));
  p = __builtin_elementwise_fma(z, p, (V)(8.504784666e-03f));
  p = __builtin_elementwise_fma(z, p, (V)(-5.395101011e-02f));
  p = __builtin_elementwise_fma(z, p, (V)(-4.584769309e-01f));
  p = __builtin_elementwise_fma(z, p, (V)(-1.151244164e+00f));
  return __builtin_elementwise_fma(z, p, (V)(-9.999961853e-01f));
}
// four values: two independent packed chains, which the scheduler interleaves (a packed fp32 instruction that consumes the result
// of the one right before it costs a wait state)
__device__ __forceinline__ void gelu_erf4(float *x) {
  const f32x4_t z = {fminf(fabsf(x[0]), 6.0f), fminf(fabsf(x[1]), 6.0f), fminf(fabsf(x[2]), 6.0f), fminf(fabsf(x[3]), 6.0f)};
  const f32x4_t p = gelu_erf_poly(z);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = fmaf(-fabsf(x[i]), __builtin_amdgcn_exp2f(p[i]), fmaxf(x[i], 0.f));
}

}  // namespace s6d
