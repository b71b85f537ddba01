// Hardware semantics of the block-scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950), as far as the MX form of the fp8 GEMM
// needs them: (1) a lane's scale byte applies to ITS 32-k block (lanes 0-31: k 0..31, lanes 32-63: k 32..63 of the same row), so two
// blocks of one row may carry different scales; (2) op_sel picks which byte of the scale VGPR is used.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mx_scale_probe.hip -o tools/probes/mx_scale_probe.bin && tools/probes/mx_scale_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OPA, int OPB>
__global__ void probe(const int *sa_words, const int *sb_words, float *out) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }          // e4m3 1.0 everywhere
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPA, sa_words[lane], OPB, sb_words[lane]);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
  int ha[64], hb[64];
  // A row r = lane & 31, block h = lane >> 5: byte j of the word = 127 + ((r + j) % 3) + 4 * h  (different per block and per byte)
  for (int l = 0; l < 64; ++l) {
    const int r = l & 31, h = l >> 5;
    unsigned w = 0;
    for (int j = 0; j < 4; ++j) w |= (unsigned)(127 + ((r + j) % 3) + 4 * h) << (8 * j);
    ha[l] = (int)w;
    unsigned v = 0;
    for (int j = 0; j < 4; ++j) v |= (unsigned)(127 - ((r + 2 * j) % 2) - 2 * h) << (8 * j);
    hb[l] = (int)v;
  }
  int *da, *db;
  float *dout, hout[64 * 16];
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(hout));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int opa = cfg, opb = (cfg * 3 + 1) & 3;
    switch (cfg) {
      case 0: hipLaunchKernelGGL((probe<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      case 1: hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      case 2: hipLaunchKernelGGL((probe<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
      default: hipLaunchKernelGGL((probe<3, 2>), dim3(1), dim3(64), 0, 0, da, db, dout); break;
    }
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    // C layout (32x32): lane l holds column n = l & 31, rows m = 8 * (i / 4) + 4 * (l >> 5) + (i % 4) ... verify against the expected
    // D[m][n] = sum over blocks h of 32 * 2^(sa(m, h) - 127) * 2^(sb(n, h) - 127) for BOTH candidate row maps and report which fits
    int bad[2] = {0, 0};
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 16; ++i) {
        const int n = l & 31;
        const int mcand[2] = {8 * (i / 4) + 4 * (l >> 5) + (i % 4), (i % 4) + 4 * (l >> 5) + 8 * (i / 4)};
        for (int q = 0; q < 2; ++q) {
          const int m = mcand[q];
          double e = 0;
          for (int h = 0; h < 2; ++h) {
            const int sa = (ha[m + 32 * h] >> (8 * opa)) & 0xff, sb = (hb[n + 32 * h] >> (8 * ((cfg == 0) ? 1 : (cfg == 1) ? 0 : (cfg == 2) ? 3 : 2))) & 0xff;
            e += 32.0 * std::ldexp(1.0, sa - 127) * std::ldexp(1.0, sb - 127);
          }
          if (std::fabs(e - hout[l * 16 + i]) > 1e-6 * e) ++bad[q];
        }
      }
    printf("op_sel_a=%d op_sel_b=%d: mismatches vs 'lane's byte[op_sel] scales its own 32-k block' = %d of 1024 (sample D[0][0..3] = %g %g %g %g)\n",
           opa, opb, bad[0], hout[0], hout[16], hout[32], hout[48]);
    bad_total += bad[0];
  }
  printf(bad_total == 0 ? "MX SEMANTICS CONFIRMED\n" : "MX SEMANTICS NOT AS ASSUMED\n");
  return 0;
}
