// hipemu -- a functional stand-in for <hip/hip_runtime.h> that lets the kernel sources of sam6d_amd/csrc compile for the HOST
// and run without a GPU (TEST INFRASTRUCTURE: tests/test_emu_*.py; nothing in the product includes this).
//
// Execution model: hipLaunchKernelGGL runs the grid one block at a time; the lanes of a block are fibers (hand-rolled stack switch) on one
// OS thread, scheduled round-robin; __syncthreads and the wave collectives (__shfl*, __ballot, MFMA, ds_read_tr16_b64) are
// rendezvous points of the block / of a 64-lane wave.  __shared__ variables are function-local statics (one block runs at a
// time); dynamic shared memory (`extern __shared__ char name[]`) is a global array the build step points the declaration at.
// MFMA and the transposing LDS read follow the gfx950 lane layouts the kernels were written against (the latter verified
// with tools/probes/trprobe.hip on the GPU).  Accumulation order inside an emulated MFMA is k-ascending in float32 -- the
// hardware's internal order is not documented, so parity tests built on this use tolerances, not bit equality, for MFMA
// results; integer / index / elementwise kernels are bit-faithful.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIPEMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 {
  unsigned x, y, z, w;
};
struct uint2 {
  unsigned x, y;
};
struct float2 {
  float x, y;
};
struct float4 {
  float x, y, z, w;
};
struct int2 {
  int x, y;
};
struct int4 {
  int x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu: no error"; }
template <class F>
static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

namespace hipemu {

struct PendingDma {                        // an LDS-DMA (global_load_lds) that has been issued but has not landed yet
  const void *src;
  void *dst;
  int bytes;
};
struct Lane {
  void *sp = nullptr;                       // saved stack pointer while the lane is switched out
  std::vector<char> stack;
  dim3 tid;
  bool done = false;
  std::vector<PendingDma> vm;               // in issue order (HIPEMU_GLDS=late only)
};

struct Block {
  std::vector<Lane> lanes;
  dim3 bid, bdim, gdim;
  int alive = 0;
  // block barrier
  int arrived = 0;
  long gen = 0;
  // per-wave rendezvous: arrival counters, generations, and 64 exchange slots of 64 bytes
  std::vector<int> w_arrived;
  std::vector<long> w_gen;
  std::vector<int> w_alive;
  std::vector<unsigned char> xchg;  // [wave][64 lanes][64 bytes]
  int cur = 0;
  std::function<void()> body;
};

extern Block *g_blk;
extern size_t g_dyn_shared_bytes;

inline Lane &self() { return g_blk->lanes[g_blk->cur]; }
inline int flat_tid() {
  const Lane &l = self();
  return (int)(l.tid.x + g_blk->bdim.x * (l.tid.y + g_blk->bdim.y * l.tid.z));
}
inline int wave_of() { return flat_tid() >> 6; }
inline int lane_of() { return flat_tid() & 63; }
void yield();
void block_barrier();
void wave_barrier();
void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
// LDS-DMA.  The hardware may land the data any time between the issue and the issuing wave's covering s_waitcnt vmcnt(N).
// HIPEMU_GLDS=early (default) lands it at the issue -- exposes a slot that is restaged while somebody still has to read it;
// HIPEMU_GLDS=late lands it only when a counted wait retires it -- exposes a read that is not covered by a wait + barrier.
// A kernel that is correct on the GPU passes under both.
void glds_issue(const void *src, void *dst_lane, int bytes);
void vmcnt_wait(int n);
inline unsigned char *slot(int lane) { return g_blk->xchg.data() + ((size_t)wave_of() * 64 + lane) * 64; }

// every lane of the wave contributes `mine`; returns after all have, with a stable view of all 64 contributions until the
// matching end_exchange() (a second rendezvous that frees the slots)
// `kind` names the collective (1 shuffle, 2 ballot, 3 MFMA bf16, 4 MFMA f32, 5 transposing LDS read): if the lanes of a wave
// meet in DIFFERENT collectives -- a collective inside divergent control flow, which on the GPU is undefined or silently wrong
// -- the run is aborted instead of exchanging unrelated operands.
void check_same_kind(int kind);
template <class T>
inline void begin_exchange(const T &mine, int kind) {
  static_assert(sizeof(T) <= 60, "exchange slot too small");
  unsigned char *s = slot(lane_of());
  std::memcpy(s, &mine, sizeof(T));
  std::memcpy(s + 60, &kind, 4);
  wave_barrier();
  if ((s - g_blk->xchg.data()) % (64 * 64) == 0) check_same_kind(kind);   // lane 0 of the wave checks once per collective
}
template <class T>
inline T peek(int lane) {
  T v;
  std::memcpy(&v, slot(lane & 63), sizeof(T));
  return v;
}
inline void end_exchange() { wave_barrier(); }

}  // namespace hipemu

#define threadIdx (hipemu::self().tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
// on the GPU a wave runs in lockstep and this is only a scheduling fence; here the lanes are independent fibers, so LDS
// traffic between lanes of one wave that is ordered by it needs a real rendezvous
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
static inline void __builtin_amdgcn_s_setprio(int) {}

template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  hipemu::begin_exchange(v, 1);
  const int me = hipemu::lane_of();
  const int src = ((me ^ mask) & (width - 1)) | (me & ~(width - 1));
  T r = hipemu::peek<T>(src);
  hipemu::end_exchange();
  return r;
}
template <class T>
static inline T __shfl(T v, int src_lane, int width = 64) {
  hipemu::begin_exchange(v, 1);
  const int me = hipemu::lane_of();
  T r = hipemu::peek<T>((src_lane & (width - 1)) | (me & ~(width - 1)));
  hipemu::end_exchange();
  return r;
}
template <class T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  hipemu::begin_exchange(v, 1);
  const int me = hipemu::lane_of();
  const int src = ((me & (width - 1)) + (int)delta < width) ? me + (int)delta : me;
  T r = hipemu::peek<T>(src);
  hipemu::end_exchange();
  return r;
}
static inline unsigned long long __ballot(int pred) {
  hipemu::begin_exchange<int>(pred != 0, 2);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (l < hipemu::g_blk->w_alive[hipemu::wave_of()] && hipemu::peek<int>(l)) m |= 1ull << l;
  hipemu::end_exchange();
  return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(!pred) == 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __brev(unsigned v) {
  unsigned r = 0;
  for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}

// v_readfirstlane_b32: the value of the first ACTIVE lane of the wave (all alive lanes here: the lanes must meet)
static inline int __builtin_amdgcn_readfirstlane(int v) {
  hipemu::begin_exchange(v, 6);
  int r = v;
  for (int l = 0; l < 64; ++l)
    if (l < hipemu::g_blk->w_alive[hipemu::wave_of()]) { r = hipemu::peek<int>(l); break; }
  hipemu::end_exchange();
  return r;
}
// v_permlane32_swap_b32 vdst, src0: lanes 32..63 of vdst <-> lanes 0..31 of src0; returns {vdst, src0}
typedef __attribute__((ext_vector_type(2))) unsigned hipemu_u32x2;
static inline hipemu_u32x2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src0, bool, bool) {
  struct P { unsigned a, b; } mine{vdst, src0};
  hipemu::begin_exchange(mine, 7);
  const int l = hipemu::lane_of();
  hipemu_u32x2 r;
  if (l < 32) { r[0] = vdst; r[1] = hipemu::peek<P>(l + 32).a; }
  else { r[0] = hipemu::peek<P>(l - 32).b; r[1] = src0; }
  hipemu::end_exchange();
  return r;
}
// v_mov_b32_dpp with a quad_perm control (dpp_ctrl < 0x100): lane l reads lane (l & ~3) + ((ctrl >> 2 (l & 3)) & 3)
static inline int __builtin_amdgcn_mov_dpp(int src, int dpp_ctrl, int, int, bool) {
  if (dpp_ctrl >= 0x100) {
    std::fprintf(stderr, "hipemu: only quad_perm DPP controls are emulated (got 0x%x)\n", dpp_ctrl);
    std::abort();
  }
  hipemu::begin_exchange(src, 10);
  const int l = hipemu::lane_of();
  const int r = hipemu::peek<int>((l & ~3) + ((dpp_ctrl >> (2 * (l & 3))) & 3));
  hipemu::end_exchange();
  return r;
}
// global_load_lds_dwordx4 (and narrower): LDS address = M0 (wave-uniform base) + lane * size; the global address is per lane
template <class G, class L>
static inline void hipemu_global_load_lds(G g, L l, int size) {
  // the destination operand goes through M0: it must be the same in every lane of the wave
  const unsigned long long base = (unsigned long long)(uintptr_t)(void *)l;
  hipemu::begin_exchange(base, 8);
  for (int k = 0; k < 64; ++k)
    if (k < hipemu::g_blk->w_alive[hipemu::wave_of()] && hipemu::peek<unsigned long long>(k) != base) {
      std::fprintf(stderr, "hipemu: global_load_lds with a lane-dependent LDS base\n");
      std::abort();
    }
  hipemu::end_exchange();
  hipemu::glds_issue((const void *)g, (char *)(void *)l + hipemu::lane_of() * size, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipemu_global_load_lds(g, l, size)

// lanes are fibers of ONE OS thread and only switch at rendezvous points: plain read-modify-write is atomic here
template <class T>
static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
static inline T atomicMax(T *p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <class T>
static inline T atomicMin(T *p, T v) { T o = *p; *p = o < v ? o : v; return o; }
template <class T>
static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T>
static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T>
static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

#define __expf(x) expf(x)   // glibc declares __expf / __logf itself
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __frsqrt_rn(float a) { return 1.0f / sqrtf(a); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __saturatef(float a) { return a < 0.f ? 0.f : (a > 1.f ? 1.f : a); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_sinf(float x) { return (float)sin(6.283185307179586476925 * (double)x); }   // input in revolutions
static inline float __builtin_amdgcn_cosf(float x) { return (float)cos(6.283185307179586476925 * (double)x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
using std::max;
using std::min;

// ---- matrix cores ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 hipemu_bf16x8;
typedef __attribute__((ext_vector_type(4))) float hipemu_f32x4;
typedef __attribute__((ext_vector_type(4))) short hipemu_s16x4;

static inline float hipemu_bf2f(__bf16 b) {
  unsigned short u;
  std::memcpy(&u, &b, 2);
  return __uint_as_float(((unsigned)u) << 16);
}

// v_mfma_f32_16x16x32_bf16: A lane l = row l%16, k (l/16)*8..+7; B lane l = column l%16, same k; C/D lane l = column l%16,
// rows (l/16)*4 + i
template <class VA, class VC>
static inline VC hipemu_mfma_16x16x32_bf16(VA a, VA b, VC c) {
  struct AB {
    hipemu_bf16x8 a, b;
  } mine;
  std::memcpy(&mine.a, &a, 16);
  std::memcpy(&mine.b, &b, 16);
  hipemu::begin_exchange(mine, 3);
  const int l = hipemu::lane_of(), col = l & 15, rb = (l >> 4) * 4;
  VC d = c;
  for (int i = 0; i < 4; ++i) {
    float acc = c[i];
    for (int k = 0; k < 32; ++k) {
      const AB ra = hipemu::peek<AB>((rb + i) + 16 * (k >> 3)), rbv = hipemu::peek<AB>(col + 16 * (k >> 3));
      acc += hipemu_bf2f(ra.a[k & 7]) * hipemu_bf2f(rbv.b[k & 7]);
    }
    d[i] = acc;
  }
  hipemu::end_exchange();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu_mfma_16x16x32_bf16(a, b, c)

// v_mfma_f32_32x32x16_bf16: A lane l = row l%32, k (l/32)*8..+7; B lane l = column l%32, same k; C/D lane l = column l%32,
// rows (r&3) + 8 (r>>2) + 4 (l/32) for register r of 16
template <class VA, class VC>
static inline VC hipemu_mfma_32x32x16_bf16(VA a, VA b, VC c) {
  struct AB {
    hipemu_bf16x8 a, b;
  } mine;
  std::memcpy(&mine.a, &a, 16);
  std::memcpy(&mine.b, &b, 16);
  hipemu::begin_exchange(mine, 9);
  const int l = hipemu::lane_of(), col = l & 31, hb = l >> 5;
  VC d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hb;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      const AB ra = hipemu::peek<AB>(row + 32 * (k >> 3)), rbv = hipemu::peek<AB>(col + 32 * (k >> 3));
      acc += hipemu_bf2f(ra.a[k & 7]) * hipemu_bf2f(rbv.b[k & 7]);
    }
    d[r] = acc;
  }
  hipemu::end_exchange();
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_32x32x16_bf16(a, b, c)

// OCP fp8 e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities, S.1111.111 = NaN, max 448
static inline float hipemu_e4m3_to_f(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 15 && m == 7) f = NAN;
  else if (e == 0) f = ldexpf((float)m, -9);
  else f = ldexpf((float)(8 + m), e - 10);
  return s ? -f : f;
}
static inline unsigned char hipemu_f_to_e4m3(float x) {      // round to nearest even, saturating at +-448
  const unsigned char sgn = std::signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (std::isnan(a)) return sgn | 0x7f;
  if (a >= 448.f) return sgn | 0x7e;
  if (a < ldexpf(1.f, -10)) return sgn;                       // below half the smallest subnormal (2^-9): 0 (tie -> even = 0)
  int e;
  frexpf(a, &e);                                              // a = f * 2^e, f in [0.5, 1)
  int E = e - 1;                                              // a = 1.xxx * 2^E
  if (E < -6) E = -6;                                         // subnormal range: fixed exponent
  const float q = ldexpf(1.f, E - 3);                         // spacing of representable values
  float r = nearbyintf(a / q) * q;                            // nearbyint: round half to even in the default mode
  if (r >= 448.f) return sgn | 0x7e;
  if (r < ldexpf(1.f, -6)) return sgn | (unsigned char)nearbyintf(r / ldexpf(1.f, -9));
  frexpf(r, &e);
  E = e - 1;
  const int m = (int)nearbyintf(r / ldexpf(1.f, E - 3)) - 8;
  return sgn | (unsigned char)(((E + 7) << 3) | m);
}
// v_cvt_pk_fp8_f32 (gfx950: OCP e4m3fn, saturating): the two results go to the low or high 16 bits of `old`
static inline int __builtin_amdgcn_cvt_pk_fp8_f32(float a, float b, int old, bool hi) {
  const unsigned pk = (unsigned)hipemu_f_to_e4m3(a) | ((unsigned)hipemu_f_to_e4m3(b) << 8);
  return hi ? (int)(((unsigned)old & 0x0000ffffu) | (pk << 16)) : (int)(((unsigned)old & 0xffff0000u) | pk);
}

// v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3 operands (cbsz = blgp = 0), as measured on the MI355X (tools/probes/
// mx_scale_probe.hip, mx_scale_probe2.hip; profiles/r04_mx_scale_probe.txt): A lane l = row l % 32 holds 32 bytes; the 64 k elements
// of a row are  k = 32 (j / 16) + 16 (l / 32) + j % 16  for byte j of lane half l / 32 -- i.e. the FIRST 16 bytes of both lane
// halves form scale block 0 (k 0..31), the second 16 bytes block 1 (k 32..63); block b is scaled by byte op_sel of the scale
// operand of the lanes of half b: value = q * 2^(byte - 127).  B alike.  C/D as the other 32x32 shapes.
typedef __attribute__((ext_vector_type(8))) int hipemu_i32x8;
template <class VC>
static inline VC hipemu_mfma_scale_32x32x64_f8(hipemu_i32x8 a, hipemu_i32x8 b, VC c, int fa, int fb, int osa, int sa, int osb, int sb) {
  if (fa != 0 || fb != 0) {
    std::fprintf(stderr, "hipemu: only fp8 e4m3 operands of the f8f6f4 MFMA are emulated\n");
    std::abort();
  }
  struct OP {
    unsigned char q[32];
    int sc;
  } mine;
  const int l = hipemu::lane_of(), col = l & 31, hb = l >> 5;
  float A[16][64];                                               // the 16 rows this lane's results need, descaled, in k order
  std::memcpy(mine.q, &a, 32);
  mine.sc = ((unsigned)sa >> (8 * (osa & 3))) & 0xff;
  hipemu::begin_exchange(mine, 11);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hb;
    const OP h0 = hipemu::peek<OP>(row), h1 = hipemu::peek<OP>(row + 32);
    for (int half = 0; half < 2; ++half) {
      const OP &o = half ? h1 : h0;
      for (int j = 0; j < 32; ++j) {
        const int blk = j >> 4;
        A[r][32 * blk + 16 * half + (j & 15)] = ldexpf(hipemu_e4m3_to_f(o.q[j]), (blk ? h1.sc : h0.sc) - 127);
      }
    }
  }
  hipemu::end_exchange();
  std::memcpy(mine.q, &b, 32);
  mine.sc = ((unsigned)sb >> (8 * (osb & 3))) & 0xff;
  hipemu::begin_exchange(mine, 12);
  float B[64];
  {
    const OP h0 = hipemu::peek<OP>(col), h1 = hipemu::peek<OP>(col + 32);
    for (int half = 0; half < 2; ++half) {
      const OP &o = half ? h1 : h0;
      for (int j = 0; j < 32; ++j) {
        const int blk = j >> 4;
        B[32 * blk + 16 * half + (j & 15)] = ldexpf(hipemu_e4m3_to_f(o.q[j]), (blk ? h1.sc : h0.sc) - 127);
      }
    }
  }
  hipemu::end_exchange();
  VC d = c;
  for (int r = 0; r < 16; ++r) {
    float acc = c[r];
    for (int k = 0; k < 64; ++k) acc += A[r][k] * B[k];
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, fa, fb, osa, sa, osb, sb) hipemu_mfma_scale_32x32x64_f8(a, b, c, fa, fb, osa, sa, osb, sb)

// IEEE-half forms of the two shapes: same lane layouts, _Float16 elements
typedef __attribute__((ext_vector_type(8))) _Float16 hipemu_f16x8;
template <class VC>
static inline VC hipemu_mfma_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, VC c) {
  struct AB {
    hipemu_f16x8 a, b;
  } mine{a, b};
  hipemu::begin_exchange(mine, 13);
  const int l = hipemu::lane_of(), col = l & 15, rb = (l >> 4) * 4;
  VC d = c;
  for (int i = 0; i < 4; ++i) {
    float acc = c[i];
    for (int k = 0; k < 32; ++k) {
      const AB ra = hipemu::peek<AB>((rb + i) + 16 * (k >> 3)), rbv = hipemu::peek<AB>(col + 16 * (k >> 3));
      acc += (float)ra.a[k & 7] * (float)rbv.b[k & 7];
    }
    d[i] = acc;
  }
  hipemu::end_exchange();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu_mfma_16x16x32_f16(a, b, c)
template <class VC>
static inline VC hipemu_mfma_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, VC c) {
  struct AB {
    hipemu_f16x8 a, b;
  } mine{a, b};
  hipemu::begin_exchange(mine, 14);
  const int l = hipemu::lane_of(), col = l & 31, hb = l >> 5;
  VC d = c;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hb;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      const AB ra = hipemu::peek<AB>(row + 32 * (k >> 3)), rbv = hipemu::peek<AB>(col + 32 * (k >> 3));
      acc += (float)ra.a[k & 7] * (float)rbv.b[k & 7];
    }
    d[r] = acc;
  }
  hipemu::end_exchange();
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu_mfma_32x32x16_f16(a, b, c)

// v_mfma_f32_16x16x4_f32: A lane l = row l%16, k = l/16; B lane l = column l%16, k = l/16; C/D as above
template <class VC>
static inline VC hipemu_mfma_16x16x4_f32(float a, float b, VC c) {
  struct AB {
    float a, b;
  } mine{a, b};
  hipemu::begin_exchange(mine, 4);
  const int l = hipemu::lane_of(), col = l & 15, rb = (l >> 4) * 4;
  VC d = c;
  for (int i = 0; i < 4; ++i) {
    float acc = c[i];
    for (int k = 0; k < 4; ++k) acc = fmaf(hipemu::peek<AB>((rb + i) + 16 * k).a, hipemu::peek<AB>(col + 16 * k).b, acc);
    d[i] = acc;
  }
  hipemu::end_exchange();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4_f32(a, b, c)

// ds_read_b64_tr_b16: every lane points at 4 consecutive 16-bit elements; inside each 16-lane group, result lane i, element j
// = element (i & 3) of what lane (4 j + (i >> 2)) pointed at  (tools/probes/trprobe.hip, gpurun_out/trprobe.txt)
template <class P>
static inline hipemu_s16x4 hipemu_ds_read_tr16_b64(P p) {
  hipemu_s16x4 mine;
  std::memcpy(&mine, (const void *)p, 8);
  hipemu::begin_exchange(mine, 5);
  const int l = hipemu::lane_of(), g = l & ~15, i = l & 15;
  hipemu_s16x4 r;
  for (int j = 0; j < 4; ++j) r[j] = hipemu::peek<hipemu_s16x4>(g + 4 * j + (i >> 2))[i & 3];
  hipemu::end_exchange();
  return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16_b64(p)

// ---- launch -------------------------------------------------------------------------------------------------------------
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                       \
  do {                                                                                    \
    (void)(stream);                                                                       \
    hipemu::run_grid(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); }); \
  } while (0)
