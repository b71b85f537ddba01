// segment_seq_sum_kernel: see s6d_misc.hip (header comment) and include/sam6d_hip.h (s6d_segment_seq_sum_f32).
// Kept in its own header, free of HIP includes, so that tests/host_cc/seqsum_host.cc can compile the SAME source for the
// host (64 std::threads + a barrier stand in for the wave) and check it against numpy without a GPU.
#pragma once

namespace s6d {

constexpr int kSeqRows = 512;                                        // rows per LDS chunk (<= 8 KB at C = 4)

__global__ __launch_bounds__(64) void segment_seq_sum_kernel(const float *__restrict__ x, const long *__restrict__ start,
                                                              const long *__restrict__ count, int C,
                                                              float *__restrict__ out) {
  __shared__ float buf[kSeqRows * 4];
  const int p = blockIdx.x, lane = threadIdx.x;
  const long n = count[p];
  const float *src = x + start[p] * C;
  float acc = 0.f;
  for (long r0 = 0; r0 < n; r0 += kSeqRows) {
    const int rows = (int)((n - r0 < kSeqRows) ? (n - r0) : kSeqRows);
    for (int i = lane; i < rows * C; i += 64) buf[i] = src[r0 * C + i];
    __syncthreads();
    if (lane < C) {
      int i = 0;
      if (r0 == 0) {                                                 // the reduction starts FROM the first row (no 0 + x0)
        acc = buf[lane];
        i = 1;
      }
      for (; i + 8 <= rows; i += 8) {                                // 8 LDS reads in flight, then the 8 adds in row order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = buf[(i + k) * C + lane];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = acc + v[k];
      }
      for (; i < rows; ++i) acc = acc + buf[i * C + lane];           // strictly in row order; nothing to contract
    }
    __syncthreads();
  }
  if (lane < C) out[(size_t)p * C + lane] = acc;
}

}  // namespace s6d
