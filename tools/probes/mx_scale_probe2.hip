// Which 32 of the 64 k elements of v_mfma_scale_f32_32x32x64_f8f6f4 does a lane's scale byte apply to?  (mx_scale_probe.hip used
// all-ones data, for which ANY split of the 64 elements into two sets of 32 gives the same sums.)  A = ones except ONE 16-byte
// group (lane half H, byte group G) = 2.0; A scale byte 127 in lanes 0-31 and 131 (2^4) in lanes 32-63; B ones, scale 127.
// all ones: D = 32 * 1 + 32 * 16 = 544;  with the 2.0 group: D = 544 + 16 * scale(group).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int H, int G>
__global__ void probe(const int *sa_words, const int *sb_words, float *out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  if (H < 2 && (lane >> 5) == H)
    for (int i = 0; i < 4; ++i) a[4 * G + i] = 0x40404040;                 // e4m3 2.0
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa_words[lane], 0, sb_words[lane]);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
int main() {
  int ha[64], hb[64];
  for (int l = 0; l < 64; ++l) { ha[l] = (l >> 5) ? 131 : 127; hb[l] = 127; }
  int *da, *db;
  float *dout, hout[64 * 16];
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(hout));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  for (int cfg = 0; cfg < 5; ++cfg) {
    switch (cfg) {
      case 0: hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      case 1: hipLaunchKernelGGL((probe<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      case 2: hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      case 3: hipLaunchKernelGGL((probe<1, 1>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      default: hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
    }
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    float mn = hout[0], mx = hout[0];
    for (int i = 0; i < 1024; ++i) { mn = hout[i] < mn ? hout[i] : mn; mx = hout[i] > mx ? hout[i] : mx; }
    if (cfg < 4) printf("2.0 in (lane half %d, bytes %2d..%2d): D = %g .. %g   -> the group's scale = %g\n", cfg >> 1, 16 * (cfg & 1), 16 * (cfg & 1) + 15, mn, mx, (mn - 544) / 16);
    else printf("all ones: D = %g .. %g (expected 544)\n", mn, mx);
  }
  return 0;
}
