// Probe of ds_read_b64_tr_b16 lane semantics on gfx950 (run on the GPU box; output documents the
// mapping used by csrc/s6d_attn.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short *out, int stride_elems) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  // lane (g,i) points at row (i>>2) of a 4-row block, 4-element chunk (i&3); group g -> next 4 rows
  const int addr = (g * 4 + (i >> 2)) * stride_elems + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int stride : {16, 104}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
