// Issue-rate probe (gfx950): wall time per wave-instruction for plain VALU, v_exp_f32, MFMA and pinned mixes at
// 1 and 2 waves/SIMD.  Sequences are inline asm so the compiler can neither pack nor reorder them; a 100 KB
// dynamic LDS request pins one workgroup per CU (256 workgroups = one per CU).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/valu_rate.bin tools/probes/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k0), "v"(k1))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define PKFMA(x2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(kk0), "v"(kk1))
#define PKADD(x2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x2) : "v"(kk1))
#define MFMA(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ void probe(float *out, int iters) {
  extern __shared__ char smem[];
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
  float k0 = 0.999f, k1 = 0.001f;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f); b[i] = (__bf16)(i * 0.5f); }
  unsigned cv = 0;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 y[8], kk0 = {0.999f, 0.999f}, kk1 = {0.001f, 0.001f};
  for (int i = 0; i < 8; ++i) y[i] = f32x2{x[2 * i], x[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) FMA(x[i]);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) EXP(x[i]);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MFMA(acc[i]);
    } else if (MODE == 3) {                            // mfma + 2 fma
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(acc[i]); FMA(x[2 * i]); FMA(x[2 * i + 1]); }
    } else if (MODE == 4) {                            // mfma + 3 fma
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(acc[i]); FMA(x[2 * i]); FMA(x[2 * i + 1]); FMA(x[(2 * i + 8) & 15]); }
    } else if (MODE == 5) {                            // mfma + exp + fma
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(acc[i]); EXP(x[2 * i]); FMA(x[2 * i + 1]); }
    } else if (MODE == 6) {                            // grouped: 8 mfma then 16 fma
#pragma unroll
      for (int i = 0; i < 8; ++i) MFMA(acc[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) FMA(x[i]);
    } else if (MODE == 7) {                            // mfma + 2 exp
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(acc[i]); EXP(x[2 * i]); EXP(x[2 * i + 1]); }
    } else if (MODE == 8) {                            // 16 max3
#pragma unroll
      for (int i = 0; i < 16; ++i) MAX3(x[i], x[(i + 1) & 15], x[(i + 2) & 15]);
    } else if (MODE == 9) {                            // 16 cvt_pk
#pragma unroll
      for (int i = 0; i < 16; ++i) { unsigned d; CVT(d, x[i], x[(i + 1) & 15]); cv ^= d; }
    } else if (MODE == 10) {                           // mfma + 6 fma (VALU-heavy mix)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        MFMA(acc[i]);
        FMA(x[2 * i]); FMA(x[2 * i + 1]); FMA(x[(2 * i + 4) & 15]); FMA(x[(2 * i + 5) & 15]); FMA(x[(2 * i + 8) & 15]);
        FMA(x[(2 * i + 9) & 15]);
      }
    } else if (MODE == 12) {                           // 16 packed fma (32 values)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) PKFMA(y[i]);
    } else if (MODE == 13) {                           // mfma + 2 packed fma
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(acc[i]); PKFMA(y[i]); PKFMA(y[(i + 4) & 7]); }
    } else if (MODE == 14) {                           // 16 packed add
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) PKADD(y[i]);
    } else if (MODE == 15) {                           // per MFMA: the softmax mix of attn_global64 today (2/3 fma, 2/3 sub, 2/3 exp, 1/3 cvt, 1/2 max)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        MFMA(acc[i]);
        if (i % 3 != 2) { FMA(x[i]); FMA(x[i + 6]); EXP(x[i + 1]); } else { FMA(x[i]); EXP(x[i + 1]); EXP(x[i + 2]); }
        if (i % 3 == 0) { unsigned d; CVT(d, x[i], x[i + 1]); cv ^= d; }
        if (i % 2 == 0) MAX3(x[i + 3], x[i + 4], x[i + 5]);
      }
    } else if (MODE == 16) {                           // the reduced mix: packed fma, no separate subtraction (1/3 pk_fma, 2/3 exp, 1/3 cvt, 1/2 max)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        MFMA(acc[i]);
        if (i % 3 == 0) PKFMA(y[i]);
        if (i % 3 != 2) EXP(x[i + 1]); else { EXP(x[i + 1]); EXP(x[i + 2]); }
        if (i % 3 == 0) { unsigned d; CVT(d, x[i], x[i + 1]); cv ^= d; }
        if (i % 2 == 0) MAX3(x[i + 3], x[i + 4], x[i + 5]);
      }
    } else if (MODE == 11) {                           // 48 fma (the VALU part of mode 10 alone)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) FMA(x[i]);
    }
  }
  float s = (float)cv;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + y[i][0] + y[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x & 15];
}

template <int MODE>
void run(const char *name, int threads, int nmfma, int nvalu) {
  float *out;
  const int blocks = 256, iters = 8192;
  (void)hipMalloc(&out, blocks * threads * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  probe<MODE><<<blocks, threads, 100 * 1024>>>(out, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<MODE><<<blocks, threads, 100 * 1024>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s waves/SIMD=%d  ns/iter(SIMD)=%8.2f   [%d mfma + %d valu per wave-iter]\n", name, threads / 256,
         ms * 1e6 / iters, nmfma, nvalu);
  (void)hipFree(out);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("16 fma", threads, 0, 16);
    run<1>("16 exp", threads, 0, 16);
    run<2>("8 mfma", threads, 8, 0);
    run<3>("8x(mfma,2fma)", threads, 8, 16);
    run<4>("8x(mfma,3fma)", threads, 8, 24);
    run<5>("8x(mfma,exp,fma)", threads, 8, 16);
    run<6>("8 mfma; 16 fma", threads, 8, 16);
    run<7>("8x(mfma,2exp)", threads, 8, 16);
    run<8>("16 max3", threads, 0, 16);
    run<9>("16 cvt_pk(+xor)", threads, 0, 32);
    run<10>("8x(mfma,6fma)", threads, 8, 48);
    run<11>("48 fma", threads, 0, 48);
    run<12>("16 pk_fma", threads, 0, 16);
    run<13>("8x(mfma,2pk_fma)", threads, 8, 16);
    run<14>("16 pk_add", threads, 0, 16);
    run<15>("6x(mfma,softmax mix)", threads, 6, 18);
    run<16>("6x(mfma,reduced mix)", threads, 6, 11);
  }
  return 0;
}
