// Fused ViT attention with decomposed relative-position bias for the SAM image encoder (gfx950).
//
// Reference: segment_anything/modeling/image_encoder.py
//   Block.forward :166-182 (norm1 -> window_partition(pad 64->70) -> attn -> unpartition),
//   Attention.forward :224-240, add_decomposed_rel_pos :325-361, get_rel_pos :292-322.
// The reference materialises (B*25*16,196,196) / (B*16,4096,4096) fp32 score tensors plus two
// partition copies per block.  Here one kernel reads the qkv GEMM output of the 4096 REAL tokens,
// forms windows by address arithmetic (out-of-image window slots take the qkv bias as q/k/v:
// quirk Q2, padded tokens are zeros after norm1 and DO act as keys), runs flash-style online
// softmax on MFMA tiles and writes the attended tokens straight into the (B,H,W,C) map.
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16, wave64):
//   S^T tile (16 keys x 16 queries) = K_tile (A: lane -> key l&15, 8 contiguous d) x Q^T (B: lane ->
//   query l&15, 8 contiguous d): the C layout then gives each lane 4 keys x 1 QUERY (col = l&15), so
//   softmax statistics are per-lane scalars (+2 shuffles across the 4 lane groups).
//   O^T tile (16 d x 16 queries)   = V^T (A, from a transposed LDS image) x P^T (B): P^T fragments
//   are exactly the exponentiated S^T registers (k-index permuted consistently on both operands),
//   so P never leaves registers and O^T keeps the per-lane query -> rescaling needs no shuffles.
//   The decomposed bias  rel_h[q,ky] + rel_w[q,kx]  comes from two per-query tables
//   T[j][q] = rel_pos[j] . q  (2S-1 entries) built with the same MFMA shape in the strip prologue.
#include "s6d_common.h"

namespace s6d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

__device__ __forceinline__ u16 f2bf(float f) {  // round-to-nearest-even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

struct AttnParams {
  const u16 *qkv;      // (B,H,W,3,nh,HD) bf16
  const u16 *qkv_bias; // (3,nh,HD) bf16  (q/k/v of out-of-image window slots)
  const u16 *rel_h;    // (2S-1,HD) bf16 or nullptr
  const u16 *rel_w;    // (2S-1,HD) bf16
  u16 *out;            // (B,H,W,nh,HD) bf16
  int B, H, W, nh;
  int ws;              // window size (0 = global attention over the H x W grid)
  int S;               // side of the attention grid (ws, or H for global)
  int T;               // tokens per attention problem = S*S
  int nwx, nwy;        // windows per image along x / y (1,1 for global)
  int LT;              // padded table length (multiple of 16, >= 2S-1)
  float scale;
};

template <int HD>
struct Cfg {
  static constexpr int KS = (HD + 31) / 32;     // k-steps of 32 over the head dim (zero padded)
  static constexpr int HDP = KS * 32;           // padded head dim
  static constexpr int DT = HD / 16;            // 16-wide d tiles of the output
  static constexpr int KROW = HDP + 8;          // K image row stride (bf16 elements): +16 B pad
  static constexpr int KT = 64;                 // keys per tile
  static constexpr int VROW = KT + 4;           // V^T image row stride per tile (bf16 elements): +8 B pad
};

// global element offset of token slot `t` of problem (b, wy, wx); returns false for an out-of-image slot
__device__ __forceinline__ bool token_offset(const AttnParams &p, int b, int wy, int wx, int t, size_t &off) {
  const int ty = t / p.S, tx = t - ty * p.S;
  const int y = (p.ws ? wy * p.ws : 0) + ty, x = (p.ws ? wx * p.ws : 0) + tx;
  off = ((size_t)(b * p.H + y) * p.W + x);
  return (y < p.H) && (x < p.W) && (t < p.T);
}

// 8 consecutive head-dim elements [d0, d0+8) of q/k/v (which = 0/1/2) for a token slot, as bf16x8
template <int HD>
__device__ __forceinline__ uint4 load_chunk(const AttnParams &p, int which, int head, bool valid, size_t tok, int d0) {
  uint4 r = make_uint4(0, 0, 0, 0);
  if (d0 >= HD) return r;
  const int C = p.nh * HD;
  const u16 *src = valid ? p.qkv + tok * (size_t)(3 * C) + (size_t)which * C + head * HD + d0
                         : p.qkv_bias + (size_t)which * C + head * HD + d0;
  return *reinterpret_cast<const uint4 *>(src);
}

// One KV tile (64 key slots, LDS resident) against one 16-query strip.
//   Kl : K image of the tile  [64][KROW]      (row = key slot, zero padded head dim)
//   Vt : V^T image of the tile [HD][VROW]     (row = d, col = key slot inside the tile)
template <int HD>
__device__ __forceinline__ void process_tile(const AttnParams &p, const u16 *Kl, const u16 *Vt, int key0,
                                             const bf16x8 (&qf)[Cfg<HD>::KS], const float *th, const float *tw,
                                             int qy, int qx, float &m_run, float &l_run,
                                             f32x4 (&oacc)[Cfg<HD>::DT], int lane) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
  float s[4][4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Kl + (sub * 16 + c) * C::KROW + ks * 32 + g * 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = key0 + sub * 16 + g * 4 + r;         // key slot of this score; query = lane & 15
      float v = acc[r] * p.scale;
      if (p.rel_h) {
        const int ky = kk / p.S, kx = kk - ky * p.S;
        const int jh = qy - ky + p.S - 1, jw = qx - kx + p.S - 1;
        const bool ok = kk < p.T;
        v += ok ? (th[c * p.LT + jh] + tw[c * p.LT + jw]) : 0.f;
      }
      s[sub][r] = kk < p.T ? v : -1e30f;
    }
  }
  // online softmax, one query per lane column (c); the 4 lane groups hold disjoint keys
  float mx = s[0][0];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[sub][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float m_new = fmaxf(m_run, mx);
  const float alpha = __expf(m_run - m_new);
  float psum = 0.f;
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[sub][r] = __expf(s[sub][r] - m_new);
      psum += s[sub][r];
    }
  psum += __shfl_xor(psum, 16);
  psum += __shfl_xor(psum, 32);
  l_run = l_run * alpha + psum;
  m_run = m_new;
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) oacc[dt] *= alpha;
  // O^T += V^T P^T : two k-steps of 32 keys; k-index e<4 -> sub 2j, e>=4 -> sub 2j+1 (same permutation on both operands)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    union { bf16x8 v; u16 h[8]; } pb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pb.h[e] = f2bf(s[2 * j][e]);
      pb.h[4 + e] = f2bf(s[2 * j + 1][e]);
    }
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      union { bf16x8 v; uint2 u[2]; } va;
      const u16 *row = Vt + (dt * 16 + c) * C::VROW + 32 * j + g * 4;
      va.u[0] = *reinterpret_cast<const uint2 *>(row);
      va.u[1] = *reinterpret_cast<const uint2 *>(row + 16);
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va.v, pb.v, oacc[dt], 0, 0, 0);
    }
  }
}

// Stage key slots [key0, key0+64) of problem (b,wy,wx,head) into the LDS images.
template <int HD, int THREADS>
__device__ __forceinline__ void stage_tile(const AttnParams &p, int b, int wy, int wx, int head, int key0, u16 *Kl,
                                           u16 *Vt, int tid) {
  using C = Cfg<HD>;
  constexpr int KPARTS = C::HDP / 8;
  for (int i = tid; i < 64 * KPARTS; i += THREADS) {
    const int key = i / KPARTS, part = i - key * KPARTS;
    size_t tok;
    const bool valid = token_offset(p, b, wy, wx, key0 + key, tok);
    *reinterpret_cast<uint4 *>(Kl + key * C::KROW + part * 8) = load_chunk<HD>(p, 1, head, valid, tok, part * 8);
  }
  constexpr int VPARTS = HD / 8;
  for (int i = tid; i < 64 * VPARTS; i += THREADS) {
    const int key = i & 63, part = i >> 6;
    size_t tok;
    const bool valid = token_offset(p, b, wy, wx, key0 + key, tok);
    union { uint4 u; u16 h[8]; } v;
    v.u = load_chunk<HD>(p, 2, head, valid, tok, part * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) Vt[(part * 8 + e) * C::VROW + key] = v.h[e];
  }
}

// Strip prologue: Q fragments, (qy,qx) of the lane's query and the two bias tables of the strip.
template <int HD>
__device__ __forceinline__ void load_strip(const AttnParams &p, int b, int wy, int wx, int head, int q0,
                                           bf16x8 (&qf)[Cfg<HD>::KS], float *th, float *tw, int lane) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
  size_t tok;
  const bool valid = token_offset(p, b, wy, wx, q0 + c, tok);
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    union { uint4 u; bf16x8 v; } x;
    x.u = load_chunk<HD>(p, 0, head, valid, tok, ks * 32 + g * 8);
    qf[ks] = x.v;
  }
  if (!p.rel_h) return;
  const int L = 2 * p.S - 1;
  for (int jt = 0; jt < p.LT / 16; ++jt) {
    f32x4 ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
    const int j = jt * 16 + c;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      union { uint4 u; bf16x8 v; } rh, rw;
      rh.u = make_uint4(0, 0, 0, 0);
      rw.u = make_uint4(0, 0, 0, 0);
      if (j < L && d0 < HD) {
        rh.u = *reinterpret_cast<const uint4 *>(p.rel_h + (size_t)j * HD + d0);
        rw.u = *reinterpret_cast<const uint4 *>(p.rel_w + (size_t)j * HD + d0);
      }
      ah = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rh.v, qf[ks], ah, 0, 0, 0);
      aw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rw.v, qf[ks], aw, 0, 0, 0);
    }
    // C layout: row j = jt*16 + g*4 + r, col = query c  ->  table[c][j]
    *reinterpret_cast<float4 *>(th + c * p.LT + jt * 16 + g * 4) = make_float4(ah[0], ah[1], ah[2], ah[3]);
    *reinterpret_cast<float4 *>(tw + c * p.LT + jt * 16 + g * 4) = make_float4(aw[0], aw[1], aw[2], aw[3]);
  }
}

template <int HD>
__device__ __forceinline__ void store_strip(const AttnParams &p, int b, int wy, int wx, int head, int q0, float l_run,
                                            const f32x4 (&oacc)[Cfg<HD>::DT], int lane) {
  using C = Cfg<HD>;
  const int g = lane >> 4, c = lane & 15;
  size_t tok;
  const bool valid = token_offset(p, b, wy, wx, q0 + c, tok);
  if (!valid) return;
  const float inv = 1.0f / l_run;
  u16 *dst = p.out + tok * (size_t)(p.nh * HD) + head * HD;
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) {
    union { uint2 u; u16 h[4]; } o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o.h[r] = f2bf(oacc[dt][r] * inv);
    *reinterpret_cast<uint2 *>(dst + dt * 16 + g * 4) = o.u;   // O^T rows g*4..g*4+3 = 4 consecutive d
  }
}

// ---- windowed: one workgroup per (image, window, head); all key slots LDS resident ------------------
template <int HD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_window_kernel(AttnParams p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntile = (p.T + 63) / 64;
  u16 *Kl = reinterpret_cast<u16 *>(smem);                         // [ntile*64][KROW]
  u16 *Vt = Kl + (size_t)ntile * 64 * C::KROW;                     // [ntile][HD][VROW]
  float *tabs = reinterpret_cast<float *>(Vt + (size_t)ntile * HD * C::VROW);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float *th = tabs + (size_t)wave * 2 * 16 * p.LT, *tw = th + 16 * p.LT;

  int id = blockIdx.x;
  const int head = id % p.nh; id /= p.nh;
  const int wx = id % p.nwx; id /= p.nwx;
  const int wy = id % p.nwy; id /= p.nwy;
  const int b = id;
  for (int t = 0; t < ntile; ++t)
    stage_tile<HD, WAVES * 64>(p, b, wy, wx, head, t * 64, Kl + (size_t)t * 64 * C::KROW, Vt + (size_t)t * HD * C::VROW, tid);
  __syncthreads();

  const int nstrip = (p.T + 15) / 16;
  for (int strip = wave; strip < nstrip; strip += WAVES) {
    const int q0 = strip * 16;
    bf16x8 qf[C::KS];
    load_strip<HD>(p, b, wy, wx, head, q0, qf, th, tw, lane);
    const int qi = q0 + (lane & 15);
    const int qy = qi / p.S, qx = qi - qy * p.S;
    float m_run = -1e30f, l_run = 0.f;
    f32x4 oacc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < ntile; ++t)
      process_tile<HD>(p, Kl + (size_t)t * 64 * C::KROW, Vt + (size_t)t * HD * C::VROW, t * 64, qf, th, tw, qy, qx,
                       m_run, l_run, oacc, lane);
    store_strip<HD>(p, b, wy, wx, head, q0, l_run, oacc, lane);
  }
}

// ---- global: one workgroup per (image, head, 64-query tile); KV tiles stream through LDS ---------------
template <int HD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_global_kernel(AttnParams p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u16 *Kl = reinterpret_cast<u16 *>(smem);                         // [64][KROW]
  u16 *Vt = Kl + (size_t)64 * C::KROW;                             // [HD][VROW]
  float *tabs = reinterpret_cast<float *>(Vt + (size_t)HD * C::VROW);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float *th = tabs + (size_t)wave * 2 * 16 * p.LT, *tw = th + 16 * p.LT;

  const int nqt = (p.T + WAVES * 16 - 1) / (WAVES * 16);
  int id = blockIdx.x;
  const int qt = id % nqt; id /= nqt;
  const int head = id % p.nh; id /= p.nh;
  const int b = id;
  const int q0 = (qt * WAVES + wave) * 16;
  bf16x8 qf[C::KS];
  load_strip<HD>(p, b, 0, 0, head, q0, qf, th, tw, lane);
  const int qi = q0 + (lane & 15);
  const int qy = qi / p.S, qx = qi - qy * p.S;
  float m_run = -1e30f, l_run = 0.f;
  f32x4 oacc[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntile = (p.T + 63) / 64;
  for (int t = 0; t < ntile; ++t) {
    __syncthreads();                                               // previous tile fully consumed
    stage_tile<HD, WAVES * 64>(p, b, 0, 0, head, t * 64, Kl, Vt, tid);
    __syncthreads();
    process_tile<HD>(p, Kl, Vt, t * 64, qf, th, tw, qy, qx, m_run, l_run, oacc, lane);
  }
  store_strip<HD>(p, b, 0, 0, head, q0, l_run, oacc, lane);
}

template <int HD>
static int launch_attn(AttnParams p, hipStream_t st) {
  using C = Cfg<HD>;
  constexpr int WAVES = 4;
  const size_t tab = (size_t)WAVES * 2 * 16 * p.LT * sizeof(float);
  if (p.ws > 0) {
    const int ntile = (p.T + 63) / 64;
    const size_t lds = (size_t)ntile * 64 * C::KROW * 2 + (size_t)ntile * HD * C::VROW * 2 + tab;
    if (lds > 160 * 1024) return S6D_EUNSUPPORTED;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_window_kernel<HD, WAVES>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned grid = (unsigned)(p.B * p.nwy * p.nwx * p.nh);
    hipLaunchKernelGGL((attn_window_kernel<HD, WAVES>), dim3(grid), dim3(WAVES * 64), lds, st, p);
  } else {
    const size_t lds = (size_t)64 * C::KROW * 2 + (size_t)HD * C::VROW * 2 + tab;
    if (lds > 160 * 1024) return S6D_EUNSUPPORTED;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_global_kernel<HD, WAVES>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nqt = (p.T + WAVES * 16 - 1) / (WAVES * 16);
    const unsigned grid = (unsigned)(p.B * p.nh * nqt);
    hipLaunchKernelGGL((attn_global_kernel<HD, WAVES>), dim3(grid), dim3(WAVES * 64), lds, st, p);
  }
  return launch_status();
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_win_attention_bf16(const void *qkv, const void *qkv_bias, const void *rel_h, const void *rel_w,
                                      int B, int H, int W, int num_heads, int head_dim, int window, float scale,
                                      void *out, void *stream) {
  if (B < 0 || H <= 0 || W <= 0 || num_heads <= 0 || window < 0) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!qkv || !qkv_bias || !out || ((rel_h == nullptr) != (rel_w == nullptr))) return S6D_EINVAL;
  if (window == 0 && H != W) return S6D_EUNSUPPORTED;
  AttnParams p;
  p.qkv = (const u16 *)qkv; p.qkv_bias = (const u16 *)qkv_bias;
  p.rel_h = (const u16 *)rel_h; p.rel_w = (const u16 *)rel_w; p.out = (u16 *)out;
  p.B = B; p.H = H; p.W = W; p.nh = num_heads; p.ws = window;
  p.S = window ? window : H; p.T = p.S * p.S;
  p.nwx = window ? (W + window - 1) / window : 1;
  p.nwy = window ? (H + window - 1) / window : 1;
  p.LT = ((2 * p.S - 1) + 15) / 16 * 16;
  p.scale = scale;
  hipStream_t st = as_stream(stream);
  switch (head_dim) {
    case 80: return launch_attn<80>(p, st);
    case 64: return launch_attn<64>(p, st);
    default: return S6D_EUNSUPPORTED;
  }
}
