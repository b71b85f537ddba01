// Row-wise epilogue arithmetic shared by plin_kernel (csrc/s6d_plin.hip) and pchain_kernel (csrc/s6d_pchain.hip): the two kernels
// must give the SAME BITS for the same layer (the fused chain replaces three launches of the other; which one runs must not show in
// a pose).  Left to the compiler, `a * b + c` is contracted into an fma or not depending on the code around it -- measured: 1 ulp
// between the two kernels on the GPU, none on the host emulator (-ffp-contract=off).  So the contraction is spelled out here, once.
#pragma once

namespace s6d {

// acc + bias, optional ReLU, + residual
__device__ __forceinline__ float pl_bias_act_res(float acc, float b, bool relu, float r) {
#pragma clang fp contract(off)
  float v = acc + b;
  if (relu) v = fmaxf(v, 0.f);
  return v + r;
}
// one term of the second LayerNorm pass: s + (v - mean)^2, as one fused multiply-add of the difference
__device__ __forceinline__ float pl_sqdev(float s, float v, float mean) {
#pragma clang fp contract(off)
  const float d = v - mean;
  return __builtin_fmaf(d, d, s);
}
// (v - mean) * rstd * gamma + beta: two roundings for the normalised value, then one fused multiply-add
__device__ __forceinline__ float pl_normalize(float v, float mean, float rstd, float g, float b) {
#pragma clang fp contract(off)
  const float t = (v - mean) * rstd;
  return __builtin_fmaf(t, g, b);
}

}  // namespace s6d
