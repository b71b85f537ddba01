// Host build of sam6d_amd/csrc/s6d_seqsum.h (tests/test_seqsum_host.py): the kernel source is compiled unchanged; the HIP
// execution model it needs is emulated -- one block at a time, its 64 lanes as std::threads, __syncthreads as a barrier,
// __shared__ storage as a static array.  Build: g++ -O2 -std=c++20 -pthread -ffp-contract=off.
#include <barrier>
#include <cstddef>
#include <thread>
#include <vector>

struct Idx3 {
  int x, y, z;
};
static thread_local Idx3 threadIdx, blockIdx;
static std::barrier<> *g_barrier = nullptr;
static inline void __syncthreads() { g_barrier->arrive_and_wait(); }
#define __global__
#define __shared__ static
#define __restrict__
#define __launch_bounds__(n)

#include "../../sam6d_amd/csrc/s6d_seqsum.h"

extern "C" void segment_seq_sum_host(const float *x, const long *start, const long *count, int P, int C, float *out) {
  for (int p = 0; p < P; ++p) {
    std::barrier<> bar(64);
    g_barrier = &bar;
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
      lanes.emplace_back([=] {
        threadIdx = {l, 0, 0};
        blockIdx = {p, 0, 0};
        s6d::segment_seq_sum_kernel(x, start, count, C, out);
      });
    for (auto &t : lanes) t.join();
  }
}
