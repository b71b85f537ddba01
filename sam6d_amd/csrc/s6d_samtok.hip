// The TOKEN side of a SAM TwoWayAttentionBlock as two kernels (gfx950; round 6).
//
// Reference: segment_anything/modeling/transformer.py:109-186 (TwoWayAttentionBlock.forward steps 1-3 for the sparse tokens) and
// :189-240 (Attention): per layer self-attention over the <= 8 prompt tokens (q / k / v / out projections, 8 heads), norm1, the
// query projection of the token->image attention, [the attention over the 4096 image tokens: s6d_samdec_tok2img_raw_bf16], its
// output projection, norm2, the 256 -> 2048 -> 256 MLP, norm3, and the k / v projections of the image->token attention that
// follows.  Rounds 2-5 ran these as ~45 library launches per layer and decoder batch (hipBLASLt on 7 x B rows, ATen elementwise /
// LayerNorm / softmax / copies): ~700 launches and ~3 ms per frame of 1024 prompts.
//
//   samtok_pre_kernel :  x = queries (+ pe);  a = SelfAttn(x, x, queries);  q1 = norm1(a | queries + a);  qp = Wq2 (q1 + pe) + b
//   samtok_post_kernel:  q2 = norm2(q1 + Wo2 att + b);  q3 = norm3(q2 + W2 relu(W1 q2 + b1) + b2);
//                        kt = Wk3 (q3 + pe) + b;  vt = Wv3 q3 + b
//
// A workgroup (256 threads) owns 4 prompts = 32 rows (8 token slots per prompt; slots >= T and prompts >= B are zero rows).
// Arithmetic = the bf16 autocast statement of the library path it replaces: every Linear multiplies bf16-rounded activations by
// bf16 weights with fp32 accumulation and rounds its result to bf16; LayerNorm, residual adds and the softmax are fp32.
// Every product is a TRANSPOSED 32-row tile product on v_mfma_f32_32x32x16_bf16 (a lane owns one token row): the weight is the A
// operand, read straight from global memory in FRAGMENT ORDER (s6d_linear_fragment_weight: 1 KiB contiguous per wave load, L2
// resident: 2.9 MB per layer) through a register ring 4 chunks deep; the activations are a bf16 row image in LDS (528-byte rows:
// conflict-free ds_read_b128).  The 2048-wide hidden layer never exists: eight 256-column slices, each expand -> ReLU -> image ->
// 256 more k of the squeeze product (csrc/s6d_pchain.hip's scheme).
#include "s6d_common.h"

namespace s6d {

typedef __attribute__((ext_vector_type(8))) __bf16 tk_bf16x8;
typedef __attribute__((ext_vector_type(16))) float tk_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned tk_u32x4;
typedef unsigned short u16;

constexpr int TK_ROWS = 32, TK_C = 256, TK_XS = 264;                // image row: 256 bf16 + 8 pad (528 B)
constexpr int TK_IMG = TK_ROWS * TK_XS;                             // elements of one image
constexpr int TK_NIMG = 5;                                          // X0, X1, Q, K, V (pre) / X0, X1 (post)
constexpr int TK_LDS_BYTES = TK_NIMG * TK_IMG * 2 + TK_ROWS * 8 * 4;

__device__ __forceinline__ u16 tk_bf16(float x) {
  union { __bf16 b; u16 u; } h;
  h.b = (__bf16)x;
  return h.u;
}
__device__ __forceinline__ float tk_f32(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float tk_round(float x) { return tk_f32(tk_bf16(x)); }

struct TkLinear {             // one nn.Linear: weight (N,K) bf16 in fragment order, bias (N) f32
  const u16 *w;
  const float *b;
};
struct TkNorm {
  const float *g, *b;
  float eps;
};
struct SamtokPre {
  const float *queries, *pe;  // (B,T,256) f32
  int B, T, add_pe;           // add_pe: 0 = the first layer (skip_first_layer_pe: q = k = v = queries, no residual), 1 = later layers
  TkLinear q, k, v, o;        // self_attn (256 -> 256 each)
  TkNorm n1;
  TkLinear q2;                // cross_attn_token_to_image.q_proj (256 -> 128)
  float *q1_out;              // (B,T,256) f32: norm1 output
  float *qp_out;              // (B,T,128) f32: projected token->image queries (bf16-rounded values); may be null
  // the token->image attention's k projection folded into the queries (s6d_samdec_tok2img_raw_bf16's operand), when wkfold != null:
  //   qfold[b, h*8 + t, c] = bf16( fold_scale * sum_d qp[b, t, 16 h + d] * Wk[16 h + d, c] ),  zero rows for t >= T
  const u16 *wkfold;          // (256,128) = Wk^T, bf16, fragment order
  float fold_scale;
  u16 *qfold_out;             // (B,64,256) bf16
};
struct SamtokPost {
  const float *q1, *att, *pe; // (B,T,256), (B,T,128), (B,T,256) f32
  int B, T;
  // att from the attention core's raw result (when y != null; att is then unused):  att[b,t,16 h + d] = sum_c y[b, h*8 + t, c] Wv[16 h + d, c] + bv
  const float *y;             // (B,64,256) f32
  const u16 *wvfold;          // (128,256) = Wv, bf16, fragment order
  const float *bvfold;        // (128) f32
  TkLinear o2;                // cross_attn_token_to_image.out_proj (128 -> 256)
  TkNorm n2;
  TkLinear l1, l2;            // mlp.lin1 (256 -> 2048), mlp.lin2 (2048 -> 256)
  TkNorm n3;
  TkLinear k3, v3;            // cross_attn_image_to_token.k_proj / v_proj (256 -> 128)
  float *q3_out, *kt_out, *vt_out;   // (B,T,256), (B,T,128), (B,T,128) f32; kt_out / vt_out may be null
  // operands of the image->token attention kernels made here instead of by library glue (each output optional):
  //   kexp[b, h*8 + t, 16 h + d] = kt[b,t,16 h + d] / 4 (block diagonal; the caller zero-fills the buffer once)
  //   k256[b, h*8 + t, c] = bf16( sum_d kexp[..] Wq[16 h + d, c] ),  cb[b, h*8 + t] = sum_d kexp[..] bq[16 h + d]
  //   vpt[b, n, h*8 + t] = bf16( sum_d vt[b,t,16 h + d] Wo[n, 16 h + d] );  slots t >= T are zero
  const u16 *wqfold;          // (256,128) = Wq^T of cross_attn_image_to_token.q_proj, bf16, fragment order
  const float *bqfold;        // (128) f32
  const u16 *wofold;          // (256,128) = Wo of cross_attn_image_to_token.out_proj, bf16, fragment order
  u16 *kexp_out;              // (B,64,128) bf16
  u16 *k256_out;              // (B,64,256) bf16
  float *cb_out;              // (B,64) f32
  u16 *vpt_out;               // (B,256,64) bf16
};

extern __shared__ __attribute__((aligned(16))) char tk_smem[];

// ---- shared pieces (one translation unit, both kernels) ---------------------------------------------------------------------
// acc[nt] += X W^T for NT tiles of 32 output channels starting at tile `tile0 + NT * wave`... the caller passes the wave's first tile.
// X: image at element offset xoff, k columns [32 kc0, 32 (kc0 + nkc)) of the image row; W: fragment order with k16 = K / 16 steps
// per tile, the steps [k16_0, k16_0 + 2 nkc) of the tiles.
template <int NT>
__device__ __forceinline__ void tk_gemm(tk_f32x16 (&acc)[NT], const u16 *__restrict__ w, int k16, int tile0, int k16_0, int nkc,
                                        const u16 *lds, int xoff, int lane) {
  constexpr int PF = 4;
  const int fr = lane & 31, fh = lane >> 5;
  const size_t t0 = ((size_t)tile0 * k16 + k16_0) * 512 + lane * 8;
  const size_t tn = (size_t)k16 * 512;
  tk_u32x4 wr[PF][2 * NT];
  auto gload = [&](int kc, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wr[s][2 * nt + ks] = *reinterpret_cast<const tk_u32x4 *>(w + t0 + nt * tn + (size_t)(2 * kc + ks) * 512);
  };
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (s < nkc) gload(s, s);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) {
    if (kc < nkc) {
      const int s = kc % PF;
      tk_bf16x8 wf[NT][2];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[nt][ks] = __builtin_bit_cast(tk_bf16x8, wr[s][2 * nt + ks]);
      if (kc + PF < nkc) gload(kc + PF, s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const tk_bf16x8 xf = *reinterpret_cast<const tk_bf16x8 *>(lds + xoff + fr * TK_XS + kc * 32 + (2 * ks + fh) * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][ks], xf, acc[nt], 0, 0, 0);
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void tk_zero(tk_f32x16 (&acc)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
}

// column of register r of tile nt for a wave whose first column is c0:  c0 + 32 nt + 4 fh + (r & 3) + 8 (r >> 2)
// acc <- round_bf16(acc + bias) (optionally ReLU first): what an autocast nn.Linear returns
template <int NT>
__device__ __forceinline__ void tk_bias(tk_f32x16 (&acc)[NT], const float *bias, int c0, int fh, bool relu) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4 *>(bias + c0 + 32 * nt + 4 * fh + 8 * q);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = tk_round(acc[nt][4 * q + e] + bv[e]);
        if (relu) v = fmaxf(v, 0.f);
        acc[nt][4 * q + e] = v;
      }
    }
}

// the lane's values -> a bf16 image (row fr, its columns), as 8-byte stores
template <int NT>
__device__ __forceinline__ void tk_to_image(const tk_f32x16 (&v)[NT], u16 *lds, int off, int c0, int fr, int fh) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned lo = (unsigned)tk_bf16(v[nt][4 * q]) | ((unsigned)tk_bf16(v[nt][4 * q + 1]) << 16);
      const unsigned hi = (unsigned)tk_bf16(v[nt][4 * q + 2]) | ((unsigned)tk_bf16(v[nt][4 * q + 3]) << 16);
      *reinterpret_cast<uint2 *>(lds + off + fr * TK_XS + c0 + 32 * nt + 4 * fh + 8 * q) = make_uint2(lo, hi);
    }
}

// LayerNorm over the 256 values of a row (2 tiles x 16 registers x 2 half-waves x 4 waves), two passes, fixed-order sums
__device__ __forceinline__ void tk_layernorm(tk_f32x16 (&acc)[2], const TkNorm &n, float (*red)[8], int c0, int wave, int fr, int fh) {
  float mean = 0.f, rstd = 0.f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[nt][r] - mean;
        s += pass ? d * d : acc[nt][r];
      }
    red[fr][2 * wave + fh] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[fr][j];
    if (pass) rstd = rsqrtf(t / 256.f + n.eps);
    else mean = t / 256.f;
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g = *reinterpret_cast<const float4 *>(n.g + c0 + 32 * nt + 4 * fh + 8 * q);
      const float4 b = *reinterpret_cast<const float4 *>(n.b + c0 + 32 * nt + 4 * fh + 8 * q);
      const float gv[4] = {g.x, g.y, g.z, g.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nt][4 * q + e] = (acc[nt][4 * q + e] - mean) * rstd * gv[e] + bv[e];
    }
}

// rows of (B,T,W) f32 tensors -> registers in the accumulator layout (the lane's row, its columns); zero for padding rows
template <int NT>
__device__ __forceinline__ void tk_load_rows(tk_f32x16 (&v)[NT], const float *src, int width, long row, bool valid, int c0, int fh) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) x = *reinterpret_cast<const float4 *>(src + row * width + c0 + 32 * nt + 4 * fh + 8 * q);
      v[nt][4 * q] = x.x; v[nt][4 * q + 1] = x.y; v[nt][4 * q + 2] = x.z; v[nt][4 * q + 3] = x.w;
    }
}
template <int NT>
__device__ __forceinline__ void tk_store_rows(const tk_f32x16 (&v)[NT], float *dst, int width, long row, bool valid, int c0, int fh) {
  if (!valid) return;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4 *>(dst + row * width + c0 + 32 * nt + 4 * fh + 8 * q) =
          make_float4(v[nt][4 * q], v[nt][4 * q + 1], v[nt][4 * q + 2], v[nt][4 * q + 3]);
}


// Per-head fold  out[c, row] = sum_{d < 16} W^T[c, 16 h + d] x[row, 16 h + d]  for the wave's 64 columns c and the 8 heads: ONE matrix
// instruction per (head, 32-column tile) -- k16 step h of the fragment-ordered (256,128) operand against columns [16 h, 16 h + 16)
// of the (32 rows x 128) bf16 image at xoff.  emit(h, nt, acc) receives the 32 x 32 tile (the lane's row, 16 of its columns).
template <typename F>
__device__ __forceinline__ void tk_head_fold(const u16 *__restrict__ w, const u16 *lds, int xoff, int wave, int lane, F emit) {
  const int fr = lane & 31, fh = lane >> 5;
  tk_u32x4 wr[2][8];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int h = 0; h < 8; ++h) wr[nt][h] = *reinterpret_cast<const tk_u32x4 *>(w + ((size_t)((2 * wave + nt) * 8 + h) * 64 + lane) * 8);
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const tk_bf16x8 xf = *reinterpret_cast<const tk_bf16x8 *>(lds + xoff + fr * TK_XS + 16 * h + 8 * fh);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      tk_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tk_bf16x8, wr[nt][h]), xf, acc, 0, 0, 0);
      emit(h, nt, acc);
    }
  }
}
// the tile of tk_head_fold as rows of a (B,64,256) bf16 tensor: out[prompt, h*8 + tok, 64 wave + 32 nt + ...] = bf16(scale * acc)
__device__ __forceinline__ void tk_store_fold(u16 *out, int prompt, int tok, bool prompt_ok, bool tok_ok, int h, int c0, int fh,
                                              const tk_f32x16 &acc, float scale) {
  if (!prompt_ok) return;
  u16 *dst = out + ((size_t)prompt * 64 + h * 8 + tok) * 256 + c0 + 4 * fh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned lo = 0, hi = 0;
    if (tok_ok) {
      lo = (unsigned)tk_bf16(acc[4 * q] * scale) | ((unsigned)tk_bf16(acc[4 * q + 1] * scale) << 16);
      hi = (unsigned)tk_bf16(acc[4 * q + 2] * scale) | ((unsigned)tk_bf16(acc[4 * q + 3] * scale) << 16);
    }
    *reinterpret_cast<uint2 *>(dst + 8 * q) = make_uint2(lo, hi);
  }
}

// ---- kernel 1: self-attention, norm1, token->image query projection --------------------------------------------------------
__global__ __launch_bounds__(256) void samtok_pre_kernel(SamtokPre p) {
  u16 *lds = reinterpret_cast<u16 *>(tk_smem);
  float(*red)[8] = reinterpret_cast<float(*)[8]>(tk_smem + TK_NIMG * TK_IMG * 2);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  constexpr int X0 = 0, X1 = TK_IMG, QI = 2 * TK_IMG, KI = 3 * TK_IMG, VI = 4 * TK_IMG;
  const int prompt = blockIdx.x * 4 + (fr >> 3), tok = fr & 7;
  const bool valid = prompt < p.B && tok < p.T;
  const long row = (long)prompt * p.T + tok;
  const int c0 = 64 * wave;                                          // the wave's columns of a 256-wide product

  // x = queries (+ pe) -> X0, queries -> X1 (bf16 images; rows loaded in the accumulator layout: each lane its row's columns)
  tk_f32x16 qv[2], pv[2];
  tk_load_rows<2>(qv, p.queries, TK_C, row, valid, c0, fh);
  tk_load_rows<2>(pv, p.pe, TK_C, row, valid, c0, fh);
  {
    tk_f32x16 x[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[nt][r] = p.add_pe ? qv[nt][r] + pv[nt][r] : qv[nt][r];
    tk_to_image<2>(x, lds, X0, c0, fr, fh);
    tk_to_image<2>(qv, lds, X1, c0, fr, fh);
  }
  __syncthreads();
  // q, k (of x), v (of queries): bf16 results -> images
  {
    tk_f32x16 a[2];
    tk_zero<2>(a);
    tk_gemm<2>(a, p.q.w, 16, 2 * wave, 0, 8, lds, X0, lane);
    tk_bias<2>(a, p.q.b, c0, fh, false);
    tk_to_image<2>(a, lds, QI, c0, fr, fh);
    tk_zero<2>(a);
    tk_gemm<2>(a, p.k.w, 16, 2 * wave, 0, 8, lds, X0, lane);
    tk_bias<2>(a, p.k.b, c0, fh, false);
    tk_to_image<2>(a, lds, KI, c0, fr, fh);
    tk_zero<2>(a);
    tk_gemm<2>(a, p.v.w, 16, 2 * wave, 0, 8, lds, X1, lane);
    tk_bias<2>(a, p.v.b, c0, fh, false);
    tk_to_image<2>(a, lds, VI, c0, fr, fh);
  }
  __syncthreads();
  // attention over the prompt's T tokens: thread = (prompt slot, head, query token); 32-wide heads (transformer.py:222-233)
  {
    const int ps = tid >> 6, h = (tid >> 3) & 7, i = tid & 7;
    const int r0 = ps * 8;
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    if (i < p.T) {
      float qf[32], s[8];
#pragma unroll
      for (int d8 = 0; d8 < 4; ++d8) {
        const tk_u32x4 v = *reinterpret_cast<const tk_u32x4 *>(lds + QI + (r0 + i) * TK_XS + h * 32 + d8 * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qf[d8 * 8 + 2 * e] = __uint_as_float(v[e] << 16);
          qf[d8 * 8 + 2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
        }
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
        if (j < p.T) {
#pragma unroll
          for (int d8 = 0; d8 < 4; ++d8) {
            const tk_u32x4 v = *reinterpret_cast<const tk_u32x4 *>(lds + KI + (r0 + j) * TK_XS + h * 32 + d8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc = __builtin_fmaf(qf[d8 * 8 + 2 * e], __uint_as_float(v[e] << 16), acc);
              acc = __builtin_fmaf(qf[d8 * 8 + 2 * e + 1], __uint_as_float(v[e] & 0xffff0000u), acc);
            }
          }
          acc = tk_round(acc) * 0.17677669529663687f;              // the bf16 score product, then .float() / sqrt(32)
          mx = fmaxf(mx, acc);
        }
        s[j] = acc;
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] = j < p.T ? __expf(s[j] - mx) : 0.f;
        sum += s[j];
      }
      const float inv = 1.f / sum;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < p.T) {
          const float pj = tk_round(s[j] * inv);                   // .to(v.dtype)
#pragma unroll
          for (int d8 = 0; d8 < 4; ++d8) {
            const tk_u32x4 v = *reinterpret_cast<const tk_u32x4 *>(lds + VI + (r0 + j) * TK_XS + h * 32 + d8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[d8 * 8 + 2 * e] = __builtin_fmaf(pj, __uint_as_float(v[e] << 16), o[d8 * 8 + 2 * e]);
              o[d8 * 8 + 2 * e + 1] = __builtin_fmaf(pj, __uint_as_float(v[e] & 0xffff0000u), o[d8 * 8 + 2 * e + 1]);
            }
          }
        }
      }
    }
    // attention output (bf16) -> X0 (its readers, the q / k products, are behind the barrier above)
#pragma unroll
    for (int d8 = 0; d8 < 4; ++d8) {
      tk_u32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (unsigned)tk_bf16(o[d8 * 8 + 2 * e]) | ((unsigned)tk_bf16(o[d8 * 8 + 2 * e + 1]) << 16);
      *reinterpret_cast<tk_u32x4 *>(lds + X0 + (r0 + i) * TK_XS + h * 32 + d8 * 8) = v;
    }
  }
  __syncthreads();
  // out_proj, residual (later layers), norm1
  tk_f32x16 q1[2];
  tk_zero<2>(q1);
  tk_gemm<2>(q1, p.o.w, 16, 2 * wave, 0, 8, lds, X0, lane);
  tk_bias<2>(q1, p.o.b, c0, fh, false);
  if (p.add_pe) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) q1[nt][r] += qv[nt][r];
  }
  tk_layernorm(q1, p.n1, red, c0, wave, fr, fh);
  tk_store_rows<2>(q1, p.q1_out, TK_C, row, valid, c0, fh);
  // token->image queries: q_proj(q1 + pe)
  {
    tk_f32x16 x[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[nt][r] = valid ? q1[nt][r] + pv[nt][r] : 0.f;
    tk_to_image<2>(x, lds, X1, c0, fr, fh);                          // (X1's reader, the v product, is behind two barriers)
  }
  __syncthreads();
  tk_f32x16 qp[1];
  tk_zero<1>(qp);
  tk_gemm<1>(qp, p.q2.w, 16, wave, 0, 8, lds, X1, lane);
  tk_bias<1>(qp, p.q2.b, 32 * wave, fh, false);
  if (p.qp_out) tk_store_rows<1>(qp, p.qp_out, 128, row, valid, 32 * wave, fh);
  if (p.wkfold) {
    // the folded queries of the attention core: qp (bf16 values) -> image (X0 columns [0, 128); its readers are behind the barriers
    // above), then one matrix instruction per head and column tile
    tk_to_image<1>(qp, lds, X0, 32 * wave, fr, fh);
    __syncthreads();
    const bool pok = prompt < p.B, tok_ok = tok < p.T;
    tk_head_fold(p.wkfold, lds, X0, wave, lane, [&](int h, int nt, const tk_f32x16 &acc) __attribute__((always_inline)) {
      tk_store_fold(p.qfold_out, prompt, tok, pok, tok_ok, h, 64 * wave + 32 * nt, fh, acc, p.fold_scale);
    });
  }
}

// ---- kernel 2: attention output projection, norm2, MLP, norm3, the image->token attention's k / v projections -----------------
__global__ __launch_bounds__(256) void samtok_post_kernel(SamtokPost p) {
  u16 *lds = reinterpret_cast<u16 *>(tk_smem);
  float(*red)[8] = reinterpret_cast<float(*)[8]>(tk_smem + TK_NIMG * TK_IMG * 2);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 31, fh = lane >> 5;
  constexpr int X0 = 0, X1 = TK_IMG;
  const int prompt = blockIdx.x * 4 + (fr >> 3), tok = fr & 7;
  const bool valid = prompt < p.B && tok < p.T;
  const long row = (long)prompt * p.T + tok;
  const int c0 = 64 * wave;

  // attention output (B,T,128) -> X0 columns [0, 128): wave w writes the 32 columns [32 w, 32 w + 32) of its rows
  {
    tk_f32x16 a[1];
    if (p.y) {
      // ... made here from the attention core's raw result: the 32 columns of wave w are heads 2 w and 2 w + 1; head h multiplies ITS
      // rows y[b, h*8 + t, :] (K = 256, fp32 -> bf16 hi + lo parts: two instructions per k step keep y to 2^-17) by W_v's 32-row tile
      // w, of which its own 16 rows are kept (registers 0-7 for the even head, 8-15 for the odd one)
      tk_f32x16 acc[2];
      tk_zero<2>(acc);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * wave + hh;
        const float *yr = p.y + ((size_t)prompt * 64 + h * 8 + tok) * 256 + 8 * fh;
        float4 yv[16][2];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          yv[ks][0] = valid ? *reinterpret_cast<const float4 *>(yr + 16 * ks) : make_float4(0.f, 0.f, 0.f, 0.f);
          yv[ks][1] = valid ? *reinterpret_cast<const float4 *>(yr + 16 * ks + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const tk_u32x4 wv = *reinterpret_cast<const tk_u32x4 *>(p.wvfold + ((size_t)(wave * 16 + ks) * 64 + lane) * 8);
          const float f[8] = {yv[ks][0].x, yv[ks][0].y, yv[ks][0].z, yv[ks][0].w, yv[ks][1].x, yv[ks][1].y, yv[ks][1].z, yv[ks][1].w};
          union { tk_bf16x8 v; u16 h[8]; } xh, xl;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh.h[e] = tk_bf16(f[e]);
            xl.h[e] = tk_bf16(f[e] - tk_f32(xh.h[e]));
          }
          acc[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tk_bf16x8, wv), xl.v, acc[hh], 0, 0, 0);
          acc[hh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tk_bf16x8, wv), xh.v, acc[hh], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(p.bvfold + 32 * wave + 4 * fh + 8 * q);
        const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) a[0][4 * q + e] = valid ? (q < 2 ? acc[0][4 * q + e] : acc[1][4 * q + e]) + bv[e] : 0.f;
      }
    } else {
      tk_load_rows<1>(a, p.att, 128, row, valid, 32 * wave, fh);
    }
    tk_to_image<1>(a, lds, X0, 32 * wave, fr, fh);
  }
  tk_f32x16 q1[2], pv[2];
  tk_load_rows<2>(q1, p.q1, TK_C, row, valid, c0, fh);
  tk_load_rows<2>(pv, p.pe, TK_C, row, valid, c0, fh);
  __syncthreads();
  tk_f32x16 q2[2];
  tk_zero<2>(q2);
  tk_gemm<2>(q2, p.o2.w, 8, 2 * wave, 0, 4, lds, X0, lane);          // K = 128
  tk_bias<2>(q2, p.o2.b, c0, fh, false);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) q2[nt][r] += q1[nt][r];
  tk_layernorm(q2, p.n2, red, c0, wave, fr, fh);                      // (its barriers also order the reads of X0 above)
  tk_to_image<2>(q2, lds, X1, c0, fr, fh);
  __syncthreads();
  // MLP: eight 256-column slices of the hidden layer
  tk_f32x16 y[2];
  tk_zero<2>(y);
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    tk_f32x16 e[2];
    tk_zero<2>(e);
    tk_gemm<2>(e, p.l1.w, 16, 8 * c + 2 * wave, 0, 8, lds, X1, lane);
    tk_bias<2>(e, p.l1.b, 256 * c + c0, fh, true);
    __syncthreads();                                                 // every wave is past the previous slice's squeeze product
    tk_to_image<2>(e, lds, X0, c0, fr, fh);
    __syncthreads();
    tk_gemm<2>(y, p.l2.w, 128, 2 * wave, 16 * c, 8, lds, X0, lane);   // K = 2048: k steps [16 c, 16 c + 16)
  }
  tk_bias<2>(y, p.l2.b, c0, fh, false);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) y[nt][r] += q2[nt][r];
  tk_layernorm(y, p.n3, red, c0, wave, fr, fh);
  tk_store_rows<2>(y, p.q3_out, TK_C, row, valid, c0, fh);
  // k_proj(q3 + pe), v_proj(q3) of the image->token attention
  {
    tk_f32x16 x[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[nt][r] = valid ? y[nt][r] + pv[nt][r] : 0.f;
    tk_to_image<2>(x, lds, X0, c0, fr, fh);                          // (X0's last readers are behind the layer norm's barriers)
    tk_to_image<2>(y, lds, X1, c0, fr, fh);
  }
  __syncthreads();
  tk_f32x16 kt[1], vt[1];
  tk_zero<1>(kt);
  tk_gemm<1>(kt, p.k3.w, 16, wave, 0, 8, lds, X0, lane);
  tk_bias<1>(kt, p.k3.b, 32 * wave, fh, false);
  if (p.kt_out) tk_store_rows<1>(kt, p.kt_out, 128, row, valid, 32 * wave, fh);
  tk_zero<1>(vt);
  tk_gemm<1>(vt, p.v3.w, 16, wave, 0, 8, lds, X1, lane);
  tk_bias<1>(vt, p.v3.b, 32 * wave, fh, false);
  if (p.vt_out) tk_store_rows<1>(vt, p.vt_out, 128, row, valid, 32 * wave, fh);
  if (!p.kexp_out && !p.k256_out && !p.vpt_out) return;
  // ---- operands of the image->token attention: scaled keys (x 1/4 = 1/sqrt(16): exact), folds with W_q^T and W_o per head ----------
  const bool pok = prompt < p.B, tok_ok = tok < p.T;
#pragma unroll
  for (int r = 0; r < 16; ++r) kt[0][r] *= 0.25f;
  __syncthreads();                                                   // every wave is past the k / v products (readers of X0 / X1)
  tk_to_image<1>(kt, lds, X0, 32 * wave, fr, fh);
  tk_to_image<1>(vt, lds, X1, 32 * wave, fr, fh);
  if (p.kexp_out && pok && tok_ok) {                                 // block diagonal: head h = 2 w + (register >> 3) keeps its 16 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = 2 * wave + (q >> 1), col = 32 * wave + 4 * fh + 8 * q;
      const unsigned lo = (unsigned)tk_bf16(kt[0][4 * q]) | ((unsigned)tk_bf16(kt[0][4 * q + 1]) << 16);
      const unsigned hi = (unsigned)tk_bf16(kt[0][4 * q + 2]) | ((unsigned)tk_bf16(kt[0][4 * q + 3]) << 16);
      *reinterpret_cast<uint2 *>(p.kexp_out + ((size_t)prompt * 64 + h * 8 + tok) * 128 + col) = make_uint2(lo, hi);
    }
  }
  if (p.cb_out) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float sacc = 0.f;
#pragma unroll
      for (int q = 2 * hh; q < 2 * hh + 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) sacc = __builtin_fmaf(kt[0][4 * q + e], p.bqfold[32 * wave + 4 * fh + 8 * q + e], sacc);
      sacc += __shfl_xor(sacc, 32);
      if (fh == 0 && pok) p.cb_out[(size_t)prompt * 64 + (2 * wave + hh) * 8 + tok] = tok_ok ? sacc : 0.f;
    }
  }
  __syncthreads();
  if (p.k256_out)
    tk_head_fold(p.wqfold, lds, X0, wave, lane, [&](int h, int nt, const tk_f32x16 &acc) __attribute__((always_inline)) {
      tk_store_fold(p.k256_out, prompt, tok, pok, tok_ok, h, 64 * wave + 32 * nt, fh, acc, 1.0f);
    });
  if (p.vpt_out)
    tk_head_fold(p.wofold, lds, X1, wave, lane, [&](int h, int nt, const tk_f32x16 &acc) __attribute__((always_inline)) {
      if (!pok) return;
      // vpt[b, n, h*8 + tok]: 2-byte stores; the eight token lanes of a prompt make a 16-byte run
      u16 *dst = p.vpt_out + ((size_t)prompt * 256 + 64 * wave + 32 * nt + 4 * fh) * 64 + h * 8 + tok;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * 64] = tok_ok ? tk_bf16(acc[r]) : (u16)0;
    });
}

}  // namespace s6d

using namespace s6d;

static bool tk_aligned(const void *a) { return (((uintptr_t)a) & 15) == 0; }

extern "C" int s6d_samdec_tokens_pre_bf16(const float *queries, const float *pe, int B, int T, int add_pe, const void *wq,
                                          const float *bq, const void *wk, const float *bk, const void *wv, const float *bv,
                                          const void *wo, const float *bo, const float *gamma1, const float *beta1, float eps1,
                                          const void *wq2, const float *bq2, float *q1_out, float *qp_out, const void *wkfold,
                                          float fold_scale, void *qfold_out, void *stream) {
  if (B < 0 || T < 1 || T > 8) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!queries || !pe || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !gamma1 || !beta1 || !wq2 || !bq2 || !q1_out)
    return S6D_EINVAL;
  if (!qp_out && !wkfold) return S6D_EINVAL;
  if (wkfold && !qfold_out) return S6D_EINVAL;
  if (!tk_aligned(queries) || !tk_aligned(pe) || !tk_aligned(q1_out) || !tk_aligned(qp_out) || !tk_aligned(wq) || !tk_aligned(wk) ||
      !tk_aligned(wv) || !tk_aligned(wo) || !tk_aligned(wq2) || !tk_aligned(wkfold) || !tk_aligned(qfold_out))
    return S6D_EINVAL;
  SamtokPre p;
  p.queries = queries; p.pe = pe; p.B = B; p.T = T; p.add_pe = add_pe ? 1 : 0;
  p.q = {(const u16 *)wq, bq}; p.k = {(const u16 *)wk, bk}; p.v = {(const u16 *)wv, bv}; p.o = {(const u16 *)wo, bo};
  p.n1 = {gamma1, beta1, eps1};
  p.q2 = {(const u16 *)wq2, bq2};
  p.q1_out = q1_out; p.qp_out = qp_out;
  p.wkfold = (const u16 *)wkfold; p.fold_scale = fold_scale; p.qfold_out = (u16 *)qfold_out;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&samtok_pre_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS_BYTES);
  hipLaunchKernelGGL(samtok_pre_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), TK_LDS_BYTES, as_stream(stream), p);
  return launch_status();
}

extern "C" int s6d_samdec_tokens_post_bf16(const float *q1, const float *att, const float *pe, int B, int T, const void *wo2,
                                           const float *bo2, const float *gamma2, const float *beta2, float eps2, const void *w1,
                                           const float *b1, const void *w2, const float *b2, const float *gamma3, const float *beta3,
                                           float eps3, const void *wk3, const float *bk3, const void *wv3, const float *bv3,
                                           float *q3_out, float *kt_out, float *vt_out, const float *y, const void *wvfold,
                                           const float *bvfold, const void *wqfold, const float *bqfold, const void *wofold,
                                           void *kexp_out, void *k256_out, float *cb_out, void *vpt_out, void *stream) {
  if (B < 0 || T < 1 || T > 8) return S6D_EINVAL;
  if (B == 0) return S6D_OK;
  if (!q1 || !pe || !wo2 || !bo2 || !gamma2 || !beta2 || !w1 || !b1 || !w2 || !b2 || !gamma3 || !beta3 || !wk3 || !bk3 || !wv3 ||
      !bv3 || !q3_out)
    return S6D_EINVAL;
  if (!att && !y) return S6D_EINVAL;
  if (y && (!wvfold || !bvfold)) return S6D_EINVAL;
  if ((k256_out || cb_out) && (!wqfold || !bqfold || !k256_out || !cb_out)) return S6D_EINVAL;
  if (vpt_out && !wofold) return S6D_EINVAL;
  if (!tk_aligned(q1) || !tk_aligned(att) || !tk_aligned(pe) || !tk_aligned(q3_out) || !tk_aligned(kt_out) || !tk_aligned(vt_out) ||
      !tk_aligned(wo2) || !tk_aligned(w1) || !tk_aligned(w2) || !tk_aligned(wk3) || !tk_aligned(wv3) || !tk_aligned(y) ||
      !tk_aligned(wvfold) || !tk_aligned(bvfold) || !tk_aligned(wqfold) || !tk_aligned(wofold) || !tk_aligned(kexp_out) ||
      !tk_aligned(k256_out) || !tk_aligned(vpt_out))
    return S6D_EINVAL;
  SamtokPost p;
  p.q1 = q1; p.att = att; p.pe = pe; p.B = B; p.T = T;
  p.o2 = {(const u16 *)wo2, bo2};
  p.n2 = {gamma2, beta2, eps2};
  p.l1 = {(const u16 *)w1, b1}; p.l2 = {(const u16 *)w2, b2};
  p.n3 = {gamma3, beta3, eps3};
  p.k3 = {(const u16 *)wk3, bk3}; p.v3 = {(const u16 *)wv3, bv3};
  p.q3_out = q3_out; p.kt_out = kt_out; p.vt_out = vt_out;
  p.y = y; p.wvfold = (const u16 *)wvfold; p.bvfold = bvfold;
  p.wqfold = (const u16 *)wqfold; p.bqfold = bqfold; p.wofold = (const u16 *)wofold;
  p.kexp_out = (u16 *)kexp_out; p.k256_out = (u16 *)k256_out; p.cb_out = cb_out; p.vpt_out = (u16 *)vpt_out;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&samtok_post_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS_BYTES);
  hipLaunchKernelGGL(samtok_post_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), TK_LDS_BYTES, as_stream(stream), p);
  return launch_status();
}
