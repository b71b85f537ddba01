// TEST INFRASTRUCTURE: C entry points over the reference's own launch wrappers (declared in its _ext_src/include/*.h next to
// at::Tensor prototypes, so they are re-declared here), for ctypes.  Host pointers; the kernels run on the emulated runtime.
#include <hip/hip_runtime.h>
#include <stdint.h>

// -fsanitize-coverage=trace-pc-guard callbacks (the kernel files only; see oracle/build_ref.py "Scheduling"): inside a launch
// every basic-block edge of kernel code hands the processor to the next thread of the block.
extern "C" void __sanitizer_cov_trace_pc_guard_init(uint32_t *start, uint32_t *stop) {
  for (uint32_t *g = start; g < stop; ++g) *g = 1;
}
extern "C" void __sanitizer_cov_trace_pc_guard(uint32_t *) {
  if (hipemu::g_blk) hipemu::yield();
}

void furthest_point_sampling_kernel_wrapper(int b, int n, int m, const float *dataset, float *temp, int *idxs);
void gather_points_kernel_wrapper(int b, int c, int n, int npoints, const float *points, const int *idx, float *out);
void query_ball_point_kernel_wrapper(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                                     int *idx);
void group_points_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out);

extern "C" {
// sampling.cpp:70-91: temp = full(1e10), idxs = zeros
void pn2ref_fps(int b, int n, int m, const float *dataset, float *temp, int *idxs) {
  furthest_point_sampling_kernel_wrapper(b, n, m, dataset, temp, idxs);
}
void pn2ref_gather(int b, int c, int n, int npoints, const float *points, const int *idx, float *out) {
  gather_points_kernel_wrapper(b, c, n, npoints, points, idx, out);
}
void pn2ref_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx) {
  query_ball_point_kernel_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx);
}
void pn2ref_group(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out) {
  group_points_kernel_wrapper(b, c, n, npoints, nsample, points, idx, out);
}
}
