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

}  // namespace s6d
