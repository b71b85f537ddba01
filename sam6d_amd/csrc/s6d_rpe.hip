// RPE attention core of the geometric transformer (gfx950).
//
// Reference: RPEMultiHeadAttention.forward, Pose_Estimation_Model/model/transformer.py:368-406
//   p = proj_p(embed)                      (B,N,N,256) -> (B,4,N,N,64)   5.09 GFLOP, 39.7 MB / call
//   s = (q k^T + einsum(q, p)) / 8 ; softmax over keys ; @ v
// Here the positional term is  q~ . e + q.b_p  with q~ = W_p^T q computed by the caller
// (see sam6d_amd/pem/layers.py), so the kernel streams the geometric embedding exactly once:
// one WAVEFRONT per query row (b,n): 64 lanes x float4 = one 256-channel row per load, the four
// head dot-products are folded with a 7-shuffle butterfly, softmax and P.V stay in the wave
// (scores in LDS).  HBM-bound on the embedding: N*256*4 B per row.
#include "s6d_common.h"

namespace s6d {

// explicit fused multiply-adds: written as a sum of products the compiler is free to contract either product of a pair, and picks
// differently per instantiation -- the instantiations (keys per trip, chosen by batch size) must agree bit for bit
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}

// RPE == false: plain multi-head attention of N query rows over M keys (cross attention of the sparse
// transformer, transformer.py:93-148): the same one-wave-per-query-row structure without the embedding stream.
// EH: the embedding is stored in IEEE half (s6d_geo_embedding_f16): 8 bytes per lane and key instead of 16, widened in registers; the
// products and sums are the float32 ones.
// KEYS: keys per trip of the score loop (1, 2 or 4; the launchers choose by the number of query rows, see rpe_keys()).
template <int WAVES, bool RPE, bool EH = false, int KEYS = 4>
__global__ __launch_bounds__(WAVES * 64) void rpe_attention_kernel(
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ qt, const float *__restrict__ qb, const void *__restrict__ embedv,
    int B, int N, int M, float scale, float *__restrict__ out, long ldq, long ldk, long ldv, long qt_bs, long qt_rs, long qt_hs,
    long qb_bs, long qb_rs, long qb_hs) {
  // qt (b, head, n, 256) at b qt_bs + n qt_rs + head qt_hs, qb (b, head, n) at b qb_bs + n qb_rs + head qb_hs: (B,4,N,256) / (B,4,N)
  // tensors of their own, or column blocks of the SAME projection output as q | k | v (round 4: W_p folded into the projection)
  // ldq / ldk / ldv: row strides (floats) of q / k / v -- 256 for contiguous tensors, the projection's row width when they are the
  // column blocks of one q | k | v (k | v) projection output (no .contiguous() copies between the Linear and the attention)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Np = (M + 3) & ~3;
  float *sc = reinterpret_cast<float *>(smem) + (size_t)wave * 4 * Np;   // [4][Np] scores of this wave
  const long row = (long)blockIdx.x * WAVES + wave;                       // b*N + n
  if (row >= (long)B * N) return;                                         // whole wave exits together
  const int b = (int)(row / N), n = (int)(row % N);
  const int g = lane >> 4;                                                // head owned by this lane group
  const int c4 = lane * 4;                                                // channels c4..c4+3 (head = c4/64 = g)

  const float4 q4 = *reinterpret_cast<const float4 *>(q + row * ldq + c4);
  float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0, t3 = t0;  // q~ of the four heads
  float qbg = 0.f;
  const float *erow = nullptr;
  const _Float16 *erow_h = nullptr;
  if (RPE) {
    const float *base = qt + (size_t)b * qt_bs + (size_t)n * qt_rs + c4;
    t0 = *reinterpret_cast<const float4 *>(base);
    t1 = *reinterpret_cast<const float4 *>(base + qt_hs);
    t2 = *reinterpret_cast<const float4 *>(base + 2 * qt_hs);
    t3 = *reinterpret_cast<const float4 *>(base + 3 * qt_hs);
    qbg = qb[(size_t)b * qb_bs + (size_t)n * qb_rs + (size_t)g * qb_hs];
    if (EH) erow_h = reinterpret_cast<const _Float16 *>(embedv) + (size_t)row * M * 256 + c4;
    else erow = reinterpret_cast<const float *>(embedv) + (size_t)row * M * 256 + c4;
  }
  const float *krow = k + (size_t)b * M * ldk + c4;
  const float *vrow = v + (size_t)b * M * ldv + c4;
  const bool hi32 = lane & 32, hi16 = lane & 16;

  // ---- scores -------------------------------------------------------------------------------
  // KEYS keys per trip (round 5): their embedding rows are KEYS independent 1-KiB loads in flight per wave and the 16-lane sums fold
  // the keys into the lanes (4 keys: 5 shuffles instead of 16).  Every sum keeps the operand pairs of the one-key form (distance 8,
  // 4, 2, 1, same partners): the same bits for every KEYS.  Measured (profiles/r05_rpe_four_keys_ab.txt): with few query rows (10
  // instances: 1970 waves on 8192 slots) one key per trip leaves the stream latency-bound -- a wave's next load waits for its own
  // seven-shuffle reduction -- and four keys run 0.173 -> 0.108 ms; with the chip full (32 instances) the extra registers cost two
  // waves per SIMD and the one-key form is 7 % faster.  The plain rows kernel (RPE = false) gains at both sizes.
  const bool hi8 = lane & 8, hi4 = lane & 4;
  for (int m0 = 0; m0 < M; m0 += KEYS) {
    float4 k4[KEYS], e4[KEYS];
#pragma unroll
    for (int u = 0; u < KEYS; ++u) {
      const int m = min(m0 + u, M - 1);                 // the last trip repeats key M - 1 (not stored)
      k4[u] = *reinterpret_cast<const float4 *>(krow + (size_t)m * ldk);
      if (RPE) {
        if (EH) {
          typedef __attribute__((ext_vector_type(4))) _Float16 h4;
          const h4 e = *reinterpret_cast<const h4 *>(erow_h + (size_t)m * 256);
          e4[u] = make_float4((float)e[0], (float)e[1], (float)e[2], (float)e[3]);
        } else {
          e4[u] = *reinterpret_cast<const float4 *>(erow + (size_t)m * 256);
        }
      }
    }
    float mine[KEYS];
#pragma unroll
    for (int u = 0; u < KEYS; ++u) {
      mine[u] = 0.f;
      if (RPE) {
        const float p0 = dot4(t0, e4[u]), p1 = dot4(t1, e4[u]), p2 = dot4(t2, e4[u]), p3 = dot4(t3, e4[u]);
        // fold 4 partials -> 1: lanes 0-31 keep heads {0,1}, 32-63 keep {2,3}; then bit 4 picks one
        float ka = hi32 ? p2 : p0, kb = hi32 ? p3 : p1;
        const float sa = hi32 ? p0 : p2, sb = hi32 ? p1 : p3;
        ka += __shfl_xor(sa, 32);
        kb += __shfl_xor(sb, 32);
        mine[u] = hi16 ? kb : ka;
        const float send = hi16 ? ka : kb;
        mine[u] += __shfl_xor(send, 16);
      }
      mine[u] += dot4(q4, k4[u]);                 // q.k for this lane's head (channels of head g)
    }
    // 16-lane sums: bit 3 of the lane picks key u & 1, bit 2 picks u >> 1 (KEYS = 4), then the plain butterfly
    float s;
    int mu = m0;
    bool writer;
    if (KEYS == 4) {
      float a0 = hi8 ? mine[1] : mine[0], a1 = hi8 ? mine[KEYS - 1] : mine[KEYS / 2];
      a0 += __shfl_xor(hi8 ? mine[0] : mine[1], 8);
      a1 += __shfl_xor(hi8 ? mine[KEYS / 2] : mine[KEYS - 1], 8);
      s = hi4 ? a1 : a0;
      s += __shfl_xor(hi4 ? a0 : a1, 4);
      mu += ((lane >> 3) & 1) + 2 * ((lane >> 2) & 1);
      writer = (lane & 3) == 0;
    } else if (KEYS == 2) {
      s = hi8 ? mine[KEYS - 1] : mine[0];
      s += __shfl_xor(hi8 ? mine[0] : mine[KEYS - 1], 8);
      s += __shfl_xor(s, 4);
      mu += (lane >> 3) & 1;
      writer = (lane & 7) == 0;
    } else {
      s = mine[0];
      s += __shfl_xor(s, 8);
      s += __shfl_xor(s, 4);
      writer = (lane & 15) == 0;
    }
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    if (writer && mu < M) sc[g * Np + mu] = (s + qbg) * scale;
  }
  __builtin_amdgcn_wave_barrier();
  // ---- softmax over keys, per head (16 lanes per head) ---------------------------------------
  float mx = -3.4e38f;
  for (int m = lane & 15; m < M; m += 16) mx = fmaxf(mx, sc[g * Np + m]);
  mx = fmaxf(mx, __shfl_xor(mx, 8));
  mx = fmaxf(mx, __shfl_xor(mx, 4));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  float sum = 0.f;
  for (int m = lane & 15; m < M; m += 16) {
    const float e = __expf(sc[g * Np + m] - mx);
    sc[g * Np + m] = e;
    sum += e;
  }
  sum += __shfl_xor(sum, 8);
  sum += __shfl_xor(sum, 4);
  sum += __shfl_xor(sum, 2);
  sum += __shfl_xor(sum, 1);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_wave_barrier();
  // ---- P.V ------------------------------------------------------------------------------------
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = 0; m < M; ++m) {
    const float a = sc[g * Np + m];
    const float4 v4 = *reinterpret_cast<const float4 *>(vrow + (size_t)m * ldv);
    // explicit fused multiply-adds: the KEYS instantiations must give the same bits, so the contraction is not left to
    // what the compiler decides around each instantiation's score loop
    acc.x = __builtin_fmaf(a, v4.x, acc.x); acc.y = __builtin_fmaf(a, v4.y, acc.y);
    acc.z = __builtin_fmaf(a, v4.z, acc.z); acc.w = __builtin_fmaf(a, v4.w, acc.w);
  }
  acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
  *reinterpret_cast<float4 *>(out + row * 256 + c4) = acc;
}

}  // namespace s6d

using namespace s6d;

// Keys per trip of the embedding stream (see the score loop): 4 up to 5000 query rows (25 instances; the chip has 8192 wave slots), else
// S6D_RPE_KEYS_FULL (1: measured 7 % faster than 4 at 32 instances; a compile-time switch for tools/probes/rpe_ab.py builds).
#ifndef S6D_RPE_KEYS_FULL
#define S6D_RPE_KEYS_FULL 1
#endif
#ifndef S6D_RPE_FOUR_KEYS_MAX_ROWS     // (0 in tests/test_emu_pose.py's variant builds: every size takes the S6D_RPE_KEYS_FULL instantiation)
#define S6D_RPE_FOUR_KEYS_MAX_ROWS 5000
#endif
static inline int rpe_keys(long rows) { return rows <= S6D_RPE_FOUR_KEYS_MAX_ROWS ? 4 : S6D_RPE_KEYS_FULL; }
#define S6D_RPE_LAUNCH(EHV, KEYSV, GRID, LDS, ST, ...)                                                              \
  do {                                                                                                            \
    if ((KEYSV) == 4) hipLaunchKernelGGL((rpe_attention_kernel<4, true, EHV, 4>), GRID, dim3(256), LDS, ST, __VA_ARGS__);      \
    else if ((KEYSV) == 2) hipLaunchKernelGGL((rpe_attention_kernel<4, true, EHV, 2>), GRID, dim3(256), LDS, ST, __VA_ARGS__); \
    else hipLaunchKernelGGL((rpe_attention_kernel<4, true, EHV, 1>), GRID, dim3(256), LDS, ST, __VA_ARGS__);                   \
  } while (0)

static int rpe_strided(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv, const float *qt, const float *qb,
                       const void *embed, bool eh, int B, int N, int C, int heads, float scale, float *out, void *stream) {
  if (B < 0 || N <= 0 || ldq < C || ldk < C || ldv < C || (ldq % 4) || (ldk % 4) || (ldv % 4)) return S6D_EINVAL;
  if (C != 256 || heads != 4) return S6D_EUNSUPPORTED;   // released model: d_model 256, 4 heads
  if (B == 0) return S6D_OK;
  if (!q || !k || !v || !qt || !qb || !embed || !out) return S6D_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return S6D_EINVAL;
  constexpr int WAVES = 4;
  const long rows = (long)B * N;
  const int Np = (N + 3) & ~3;
  const size_t lds = (size_t)WAVES * 4 * Np * sizeof(float);
  if (lds > 64 * 1024) return S6D_EUNSUPPORTED;
  static_assert(WAVES == 4, "S6D_RPE_LAUNCH instantiates the four-wave kernels");
  const dim3 grid((unsigned)((rows + WAVES - 1) / WAVES));
  const int keys = rpe_keys(rows);
  if (eh)
    S6D_RPE_LAUNCH(true, keys, grid, lds, as_stream(stream), q, k, v, qt, qb, embed, B, N, N, scale, out, ldq, ldk, ldv, 4L * N * 256,
                   256L, (long)N * 256, 4L * N, 1L, (long)N);
  else
    S6D_RPE_LAUNCH(false, keys, grid, lds, as_stream(stream), q, k, v, qt, qb, embed, B, N, N, scale, out, ldq, ldk, ldv, 4L * N * 256,
                   256L, (long)N * 256, 4L * N, 1L, (long)N);
  return launch_status();
}

extern "C" int s6d_rpe_attention_strided_f32(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv,
                                             const float *qt, const float *qb, const float *embed, int B, int N, int C, int heads,
                                             float scale, float *out, void *stream) {
  return rpe_strided(q, ldq, k, ldk, v, ldv, qt, qb, embed, false, B, N, C, heads, scale, out, stream);
}

extern "C" int s6d_rpe_attention_strided_e16_f32(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv,
                                                 const float *qt, const float *qb, const void *embed_f16, int B, int N, int C, int heads,
                                                 float scale, float *out, void *stream) {
  return rpe_strided(q, ldq, k, ldk, v, ldv, qt, qb, embed_f16, true, B, N, C, heads, scale, out, stream);
}

// q | k | v | q~ (4 x 256) | qb (4) as column blocks of ONE projection output proj (B,N,ld): W_p of the RPE layer is folded into the
// projection's weights by the caller (q~_h = x (W_q,h^T W_p,h) + b_q,h W_p,h), so the `W_p^T q` products of the layer are not a pass.
static int rpe_packed(const float *proj, long ld, int q_off, int k_off, int v_off, int qt_off, int qb_off, const void *embed, bool eh,
                      int B, int N, int C, int heads, float scale, float *out, void *stream) {
  if (B < 0 || N <= 0 || C != 256 || heads != 4 || (ld % 4) != 0) return B < 0 || N <= 0 || (ld % 4) ? S6D_EINVAL : S6D_EUNSUPPORTED;
  if (q_off < 0 || k_off < 0 || v_off < 0 || qt_off < 0 || qb_off < 0 || (q_off % 4) || (k_off % 4) || (v_off % 4) || (qt_off % 4) ||
      q_off + C > ld || k_off + C > ld || v_off + C > ld || qt_off + 4 * C > ld || qb_off + 4 > ld)
    return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!proj || !embed || !out || ((uintptr_t)proj & 15)) return S6D_EINVAL;
  constexpr int WAVES = 4;
  const long rows = (long)B * N;
  const int Np = (N + 3) & ~3;
  const size_t lds = (size_t)WAVES * 4 * Np * sizeof(float);
  if (lds > 64 * 1024) return S6D_EUNSUPPORTED;
  static_assert(WAVES == 4, "S6D_RPE_LAUNCH instantiates the four-wave kernels");
  const dim3 grid((unsigned)((rows + WAVES - 1) / WAVES));
  const int keys = rpe_keys(rows);
  if (eh)
    S6D_RPE_LAUNCH(true, keys, grid, lds, as_stream(stream), proj + q_off, proj + k_off, proj + v_off, proj + qt_off, proj + qb_off,
                   embed, B, N, N, scale, out, ld, ld, ld, (long)N * ld, ld, 256L, (long)N * ld, ld, 1L);
  else
    S6D_RPE_LAUNCH(false, keys, grid, lds, as_stream(stream), proj + q_off, proj + k_off, proj + v_off, proj + qt_off, proj + qb_off,
                   embed, B, N, N, scale, out, ld, ld, ld, (long)N * ld, ld, 256L, (long)N * ld, ld, 1L);
  return launch_status();
}

extern "C" int s6d_rpe_attention_packed_f32(const float *proj, long ld, int q_off, int k_off, int v_off, int qt_off, int qb_off,
                                            const float *embed, int B, int N, int C, int heads, float scale, float *out,
                                            void *stream) {
  return rpe_packed(proj, ld, q_off, k_off, v_off, qt_off, qb_off, embed, false, B, N, C, heads, scale, out, stream);
}

extern "C" int s6d_rpe_attention_packed_e16_f32(const float *proj, long ld, int q_off, int k_off, int v_off, int qt_off, int qb_off,
                                                const void *embed_f16, int B, int N, int C, int heads, float scale, float *out,
                                                void *stream) {
  return rpe_packed(proj, ld, q_off, k_off, v_off, qt_off, qb_off, embed_f16, true, B, N, C, heads, scale, out, stream);
}

extern "C" int s6d_rpe_attention_f32(const float *q, const float *k, const float *v, const float *qt,
                                     const float *qb, const float *embed, int B, int N, int C, int heads,
                                     float scale, float *out, void *stream) {
  return s6d_rpe_attention_strided_f32(q, C, k, C, v, C, qt, qb, embed, B, N, C, heads, scale, out, stream);
}

extern "C" int s6d_mha_strided_f32(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv, int B, int N, int M,
                                   int C, int heads, float scale, float *out, void *stream) {
  if (B < 0 || N <= 0 || M <= 0 || ldq < C || ldk < C || ldv < C || (ldq % 4) || (ldk % 4) || (ldv % 4)) return S6D_EINVAL;
  if (C != 256 || heads != 4) return S6D_EUNSUPPORTED;
  if (B == 0) return S6D_OK;
  if (!q || !k || !v || !out) return S6D_EINVAL;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return S6D_EINVAL;
  constexpr int WAVES = 4;
  const long rows = (long)B * N;
  const int Np = (M + 3) & ~3;
  const size_t lds = (size_t)WAVES * 4 * Np * sizeof(float);
  if (lds > 64 * 1024) return S6D_EUNSUPPORTED;
  hipLaunchKernelGGL((rpe_attention_kernel<WAVES, false, false, 4>), dim3((unsigned)((rows + WAVES - 1) / WAVES)), dim3(WAVES * 64),
                     lds, as_stream(stream), q, k, v, nullptr, nullptr, nullptr, B, N, M, scale, out, ldq, ldk, ldv, 0L, 0L, 0L, 0L, 0L, 0L);
  return launch_status();
}

extern "C" int s6d_mha_f32(const float *q, const float *k, const float *v, int B, int N, int M, int C, int heads,
                           float scale, float *out, void *stream) {
  return s6d_mha_strided_f32(q, C, k, C, v, C, B, N, M, C, heads, scale, out, stream);
}

// Focused-linear-attention feature map (LinearAttention.forward, transformer.py:536-547), one pass:
//   t = (relu(x) + 1e-6) / softplus(scale);  y = t^p / |t^p| * |t|      (p = focusing_factor = 3)
// The reference spends ~9 element-wise / reduction passes over (B,2048,256) per call.  One wave per row.
__global__ __launch_bounds__(256) void focus_kernel(const float *__restrict__ x, const float *__restrict__ inv_scale,
                                                    long rows, int p, float *__restrict__ y) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 a = *reinterpret_cast<const float4 *>(x + row * 256 + lane * 4);
  const float4 is = *reinterpret_cast<const float4 *>(inv_scale + lane * 4);
  float t[4] = {(fmaxf(a.x, 0.f) + 1e-6f) * is.x, (fmaxf(a.y, 0.f) + 1e-6f) * is.y, (fmaxf(a.z, 0.f) + 1e-6f) * is.z,
                (fmaxf(a.w, 0.f) + 1e-6f) * is.w};
  float n1 = t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
  float u[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float r = t[e];
    for (int i = 1; i < p; ++i) r *= t[e];
    u[e] = r;
  }
  float n3 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3];
  n1 = wave_sum(n1);
  n3 = wave_sum(n3);
  const float f = sqrtf(n1) / sqrtf(n3);
  *reinterpret_cast<float4 *>(y + row * 256 + lane * 4) = make_float4(u[0] * f, u[1] * f, u[2] * f, u[3] * f);
}

extern "C" int s6d_linear_attn_focus_f32(const float *x, const float *inv_scale, long rows, int C, int power, float *y,
                                         void *stream) {
  if (rows < 0 || power < 1) return S6D_EINVAL;
  if (C != 256) return S6D_EUNSUPPORTED;
  if (rows == 0) return S6D_OK;
  if (!x || !inv_scale || !y) return S6D_EINVAL;
  hipLaunchKernelGGL(focus_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, inv_scale, rows,
                     power, y);
  return launch_status();
}
