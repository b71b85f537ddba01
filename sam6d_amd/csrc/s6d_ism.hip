// Proposal-vs-template scoring kernels of the Instance Segmentation Model (gfx950).
//
// Reference (Instance_Segmentation_Model/):
//   model/loss.py:27-44    PairwiseSimilarity  -- Python loop over objects on a (P,O,T,C) repeat
//   model/loss.py:52-76    MaskedPatch_MatrixSimilarity.compute_straight / compute_visible_ratio
//                          -- the SAME (S,256,256) batched GEMM computed twice
//   model/detector.py:209-246, utils/trimesh_utils.py:77-105  masked-depth mean translation on a
//                          (S,H,W) repeat, template projection, bbox
// All GEMM-shaped work uses the exact-f32 MFMA (v_mfma_f32_16x16x4_f32: an fma chain, same numerics as
// a scalar loop) so scores keep fp32 parity with the reference; reductions (row-max, col-max, norms,
// non-zero counts) are epilogues of the tile that produced them -- the similarity matrices never reach HBM.
#include "s6d_common.h"

namespace s6d {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------------------------------------
// cosine(P queries, R references), clamp to [0,1].  One wave per 16x16 output tile.
// MFMA 16x16x4 f32: A lane -> (row l&15, k l>>4), B lane -> (k l>>4, col l&15), C: col l&15, row (l>>4)*4+r.
// Each lane loads float4 (4 consecutive k) and feeds element e to MFMA step e: the k-permutation is the
// same on both operands, so the contraction is unchanged.
__global__ __launch_bounds__(256) void pairwise_cosine_kernel(const float *__restrict__ q, const float *__restrict__ ref,
                                                             int P, int R, int C, float *__restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles_r = (R + 15) / 16;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ((P + 15) / 16) * tiles_r) return;
  const int p0 = (tile / tiles_r) * 16, r0 = (tile % tiles_r) * 16;
  const int c = lane & 15, g = lane >> 4;
  const float *qa = q + (size_t)min(p0 + c, P - 1) * C + g * 4;
  const float *rb = ref + (size_t)min(r0 + c, R - 1) * C + g * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float nq = 0.f, nr = 0.f;
  for (int k0 = 0; k0 < C; k0 += 16) {
    const float4 a = *reinterpret_cast<const float4 *>(qa + k0);
    const float4 b = *reinterpret_cast<const float4 *>(rb + k0);
    nq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    nr += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  nq += __shfl_xor(nq, 16); nq += __shfl_xor(nq, 32);      // |q_row|^2 in every lane with l&15 == row
  nr += __shfl_xor(nr, 16); nr += __shfl_xor(nr, 32);      // |ref_col|^2 in every lane with l&15 == col
  const float inv_r = 1.0f / fmaxf(sqrtf(nr), 1e-12f);     // F.normalize eps (loss.py:32-33)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = g * 4 + r;
    const float nqr = __shfl(nq, row);
    float v = acc[r] * (1.0f / fmaxf(sqrtf(nqr), 1e-12f)) * inv_r;
    v = fminf(fmaxf(v, 0.f), 1.f);
    if (p0 + row < P && r0 + c < R) out[(size_t)(p0 + row) * R + r0 + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Per proposal p: avg of the top-K template scores per object, best object, its score and its best template
// (detector.py:260-296 with aggregation 'avg_5', best_template_pose :198-207).  One wave per proposal.
__global__ __launch_bounds__(256) void semantic_select_kernel(const float *__restrict__ scores, int P, int O, int T,
                                                             int topk, float *__restrict__ best_score,
                                                             int *__restrict__ best_obj, int *__restrict__ best_tmpl) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= P) return;
  float bscore = -1.f;
  int bobj = 0, btmpl = 0;
  for (int o = 0; o < O; ++o) {
    const float *s = scores + ((size_t)p * O + o) * T;
    // lane-strided copy of the row (T <= 64*4 handled by 4 slots per lane)
    float v[4];
    int vi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = lane + 64 * j;
      v[j] = t < T ? s[t] : -2.f;
      vi[j] = t;
    }
    float sum = 0.f;
    int first_t = 0;
    for (int k = 0; k < topk && k < T; ++k) {
      // wave arg-max (value desc, index asc) over the remaining entries
      float m = v[0];
      int mi = vi[0];
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (v[j] > m) { m = v[j]; mi = vi[j]; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off);
        const int oi = __shfl_xor(mi, off);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
      }
      sum += m;
      if (k == 0) first_t = mi;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (vi[j] == mi) v[j] = -3.f;                       // knock out the selected entry
    }
    const float avg = sum / (float)(topk < T ? topk : T);
    if (avg > bscore) { bscore = avg; bobj = o; btmpl = first_t; }   // first maximum wins, like torch.max
  }
  if (lane == 0) {
    best_score[p] = bscore;
    best_obj[p] = bobj;
    best_tmpl[p] = btmpl;
  }
}

// ------------------------------------------------------------------------------------------------
// Appearance score + visible ratio from ONE pass over sim = Q_s R_s^T  (Q_s: N1 x C, R_s: N2 x C, fp32).
// The product runs on the bf16 matrix cores with a 3-term split (x = x_hi + x_lo in bf16; hi*hi + lo*hi + hi*lo,
// fp32 accumulate: ~2^-17 relative per product, |error| ~ 3e-7 on these unit-vector cosines, exact zeros stay
// exact) -- 5x the rate of the fp32 MFMA the first version used.  Workgroup = (proposal s, 256-row block of Q_s),
// 8 waves; a wave owns 32 query rows (two 16-row strips sharing every B fragment) x all N2 (<= 256) reference
// patches = 32 accumulator tiles.  Per 32-channel k-step the workgroup converts the reference slice to hi / lo bf16
// ONCE into LDS (double buffered; the next slice's global loads fly under the MFMAs), every wave converts its own
// A fragments in registers.  Epilogue per wave: row maxima summed, column maxima, and the number of query rows
// whose element sum is non-zero -- written as partials, folded by patch_finalize.
constexpr int kMaxColTiles = 16;
constexpr int kPsRow = 40;                 // LDS row stride (bf16): 32 channels + 8 pad (80 B: conflict-free b128 reads)
constexpr int kPsThreads = 512;
constexpr int kPsRowsPerWave = 32;
constexpr int kPsLdsBytes = 2 * 2 * 256 * kPsRow * 2;   // [buffer][hi | lo][256 rows][kPsRow] bf16 = 80 KB

typedef __attribute__((ext_vector_type(8))) __bf16 ps_bf16x8;

__device__ __forceinline__ void ps_split(float x, unsigned short &hi, unsigned short &lo) {
  union { __bf16 b; unsigned short u; } h, l;
  h.b = (__bf16)x;
  l.b = (__bf16)(x - __uint_as_float(((unsigned)h.u) << 16));
  hi = h.u;
  lo = l.u;
}

__global__ __launch_bounds__(kPsThreads) void patch_scores_kernel(const float *__restrict__ q,
                                                                 const float *__restrict__ refstore,
                                                                 const int *__restrict__ obj, const int *__restrict__ tmpl,
                                                                 const int *__restrict__ qsel, int N1, int N2, int C, int T,
                                                                 float *__restrict__ part_rowsum,
                                                                 float *__restrict__ part_colmax,
                                                                 int *__restrict__ part_nnz) {
  extern __shared__ __attribute__((aligned(16))) char ps_smem[];
  unsigned short *lds = reinterpret_cast<unsigned short *>(ps_smem);
  const int s = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 15, g = lane >> 4;
  const int nslot = gridDim.x * 8;
  const int slot = blockIdx.x * 8 + wave;                  // partial index within the proposal
  const int row0 = slot * kPsRowsPerWave;
  const int nct = (N2 + 15) / 16;
  const float *Q = q + (size_t)(qsel ? qsel[s] : s) * N1 * C;       // qsel: rows of the un-gathered (P,N1,C) query tensor
  const float *Rf = refstore + ((size_t)obj[s] * T + tmpl[s]) * (size_t)N2 * C;
  const float *qa[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) qa[n] = Q + (size_t)min(row0 + n * 16 + c, N1 - 1) * C + g * 8;
  // staging of the reference slice: 256 rows x 32 channels = 2048 float4, 4 per thread (rows past N2 re-read the
  // last row; their columns are masked in the epilogue)
  const float *bp[4];
  int bo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int it = tid + j * kPsThreads, row = it >> 3, c4 = it & 7;
    bp[j] = Rf + (size_t)min(row, N2 - 1) * C + c4 * 4;
    bo[j] = row * kPsRow + c4 * 4;
  }
  float4 breg[4], areg[2][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) breg[j] = *reinterpret_cast<const float4 *>(bp[j] + k0);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      areg[n][0] = *reinterpret_cast<const float4 *>(qa[n] + k0);
      areg[n][1] = *reinterpret_cast<const float4 *>(qa[n] + k0 + 4);
    }
  };
  auto bstore = [&](int buf) {
    unsigned short *hi = lds + (size_t)buf * 2 * 256 * kPsRow, *lo = hi + 256 * kPsRow;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      union { uint2 u; unsigned short h[4]; } vh, vl;
      const float v[4] = {breg[j].x, breg[j].y, breg[j].z, breg[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) ps_split(v[e], vh.h[e], vl.h[e]);
      *reinterpret_cast<uint2 *>(hi + bo[j]) = vh.u;
      *reinterpret_cast<uint2 *>(lo + bo[j]) = vl.u;
    }
  };
  f32x4 acc[2][kMaxColTiles];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int t = 0; t < kMaxColTiles; ++t) acc[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float qsum[2] = {0.f, 0.f};
  gload(0);
  bstore(0);
  __syncthreads();
  const int nk = C / 32;
  for (int ks = 0; ks < nk; ++ks) {
    // this step's A fragments (registers), then the next slice's loads take off under the MFMAs
    union { ps_bf16x8 v; unsigned short h[8]; } ah[2], al[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float v[8] = {areg[n][0].x, areg[n][0].y, areg[n][0].z, areg[n][0].w,
                          areg[n][1].x, areg[n][1].y, areg[n][1].z, areg[n][1].w};
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ps_split(v[e], ah[n].h[e], al[n].h[e]);
        sum += v[e];
      }
      qsum[n] += sum;
    }
    gload(min(ks + 1, nk - 1) * 32);                         // the last iteration re-loads in-bounds data (unused)
    const unsigned short *hi = lds + (size_t)(ks & 1) * 2 * 256 * kPsRow, *lo = hi + 256 * kPsRow;
#pragma unroll
    for (int t = 0; t < kMaxColTiles; ++t) {
      const int o = (t * 16 + c) * kPsRow + g * 8;
      const ps_bf16x8 bh = *reinterpret_cast<const ps_bf16x8 *>(hi + o);
      const ps_bf16x8 bl = *reinterpret_cast<const ps_bf16x8 *>(lo + o);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        acc[n][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[n].v, bh, acc[n][t], 0, 0, 0);
        acc[n][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[n].v, bh, acc[n][t], 0, 0, 0);
        acc[n][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[n].v, bl, acc[n][t], 0, 0, 0);
      }
    }
    if (ks + 1 < nk) bstore((ks + 1) & 1);                   // the other buffer: last read in iteration ks-1
    __syncthreads();
  }
  // ---- epilogue: A = query rows, B = reference rows -> C layout row = query g*4+r, col = reference c -----------
  int nnz = 0;
  float rsum = 0.f;
  float cmax[kMaxColTiles];
#pragma unroll
  for (int t = 0; t < kMaxColTiles; ++t) cmax[t] = -3.4e38f;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r0 = row0 + n * 16;
    float qs = qsum[n];
    qs += __shfl_xor(qs, 16);
    qs += __shfl_xor(qs, 32);                               // element sum of query row r0 + c
    nnz += __popcll(__ballot(r0 + c < N1 && qs != 0.f && lane < 16));
#pragma unroll
    for (int r = 0; r < 4; ++r) {                           // row maxima: max over column tiles and the group's 16 lanes
      float m = -3.4e38f;
#pragma unroll
      for (int t = 0; t < kMaxColTiles; ++t)
        if (t < nct && t * 16 + c < N2) m = fmaxf(m, acc[n][t][r]);
      m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2));
      m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
      if (r0 + g * 4 + r < N1) rsum += m;
    }
#pragma unroll
    for (int t = 0; t < kMaxColTiles; ++t)                  // column maxima over this strip's 16 rows
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r0 + g * 4 + r < N1) cmax[t] = fmaxf(cmax[t], acc[n][t][r]);
  }
  rsum += __shfl_xor(rsum, 16); rsum += __shfl_xor(rsum, 32);   // all 32 rows of the wave (lanes of a group agree)
#pragma unroll
  for (int t = 0; t < kMaxColTiles; ++t) {
    if (t < nct) {
      float m = cmax[t];
      m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
      if (lane < 16 && t * 16 + c < N2) part_colmax[((size_t)s * nslot + slot) * N2 + t * 16 + c] = m;
    }
  }
  if (lane == 0) {
    part_rowsum[(size_t)s * nslot + slot] = rsum;
    part_nnz[(size_t)s * nslot + slot] = nnz;
  }
}

__global__ void patch_finalize_kernel(const float *__restrict__ part_rowsum, const float *__restrict__ part_colmax,
                                      const int *__restrict__ part_nnz, int S, int nslot, int N1, int N2, float thred,
                                      float *__restrict__ appe, float *__restrict__ ratio) {
  const int s = blockIdx.x, lane = threadIdx.x;            // one wave per proposal
  float rs = 0.f;
  int nz = 0;
  for (int i = 0; i < nslot; ++i) {
    const int row0 = i * kPsRowsPerWave;
    if (row0 < N1) {
      rs += part_rowsum[(size_t)s * nslot + i];
      nz += part_nnz[(size_t)s * nslot + i];
    }
  }
  int valid = 0, vis = 0;
  for (int j = lane; j < N2; j += 64) {
    float m = -3.4e38f;
    for (int i = 0; i < nslot; ++i)
      if (i * kPsRowsPerWave < N1) m = fmaxf(m, part_colmax[((size_t)s * nslot + i) * N2 + j]);
    valid += (m != 0.f);
    vis += (m > thred) && (m != 0.f);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    valid += __shfl_xor(valid, off);
    vis += __shfl_xor(vis, off);
  }
  if (lane == 0) {
    appe[s] = fminf(fmaxf(rs / ((float)nz + 1e-6f), 0.f), 1.f);
    ratio[s] = (float)vis / ((float)valid + 1e-6f);
  }
}

// ------------------------------------------------------------------------------------------------
// Mean back-projected point of every mask (detector.py:234-246 + trimesh_utils.py:77-105) -- three masked
// sums, no (S,H,W) depth repeat.  Dtype trail of the reference: Z float32; X, Y float64 (the camera matrix is a
// float64 tensor); the three means are cast to float32 at the end.
//
// SUMMATION ORDER (round 3).  The projected pixels downstream are truncated floats, so the translation has to carry the
// reference's bits, and a float32 sum over ~10^4 pixels carries its order.  The pinned run is ATen's CPU `sum`
// (cascade_sum / vectorized_inner_sum, aten/src/ATen/native/cpu/SumKernel.cpp): the flat H*W map is read as rows of
// C = 4*V elements (V = 8 floats / 4 doubles: the AVX2 vector the kernel is compiled for, also on AVX-512 hosts), each of
// the C columns is summed on its own through a cascade -- blocks of 16 rows summed in row order from zero, 16 block sums
// added in order into a level-1 node, 16 of those into a level-2 node, level-2 nodes accumulated in order -- then the row
// remainder, the partial level-1 / level-2 nodes and the level-3 accumulator are combined in that order, the four vectors
// of a row are added 1, 2, 3 onto vector 0, and the V lanes are added in lane order onto the scalar tail.  That is a tree
// with fixed shape: level-0 blocks and level-1 nodes are independent (kernel 1: one workgroup per 8192-element span = one
// float32 level-1 node = two float64 level-1 nodes), the upper levels are a few dozen sequential adds per column
// (kernel 2: one wave per mask).  oracle/aten_sum.py restates the same order in numpy; both reproduce torch.sum bit for bit.
constexpr int kMdSpan = 8192;          // elements per workgroup of kernel 1
constexpr int kMdC32 = 32, kMdC64 = 16;   // columns of the float32 / float64 reductions (4 vectors of 8 / 4 lanes)

struct MdGeom {
  int n;            // H * W
  int nspan;        // ceil(n / kMdSpan)
  int nb32, nb64;   // complete level-0 blocks (16 rows) of the float32 / float64 reductions
};

__host__ __device__ inline MdGeom md_geom(int H, int W) {
  MdGeom g;
  g.n = H * W;
  g.nspan = (g.n + kMdSpan - 1) / kMdSpan;
  g.nb32 = ((g.n / 8) / 4) / 16;
  g.nb64 = ((g.n / 4) / 4) / 16;
  return g;
}

// per-mask workspace: [nspan][32] f32 level-1 sums of Z | [2][2*nspan][16] f64 level-1 sums of X, Y | [nspan] i32 counts
__host__ __device__ inline size_t md_ws_f64_off(const MdGeom &g) { return (((size_t)g.nspan * kMdC32 * 4) + 7) & ~(size_t)7; }
__host__ __device__ inline size_t md_ws_cnt_off(const MdGeom &g) { return md_ws_f64_off(g) + (size_t)2 * 2 * g.nspan * kMdC64 * 8; }
__host__ __device__ inline size_t md_ws_per_mask(const MdGeom &g) { return (md_ws_cnt_off(g) + (size_t)g.nspan * 4 + 15) & ~(size_t)15; }

__device__ __forceinline__ float md_z(float m, float d, float depth_scale) { return m * d * depth_scale / 1000.f; }

__global__ __launch_bounds__(256) void masked_depth_l1_kernel(const float *__restrict__ masks, const float *__restrict__ depth,
                                                             int H, int W, float depth_scale,
                                                             const double *__restrict__ K, const int *__restrict__ frame,
                                                             const int *__restrict__ msel, char *__restrict__ ws) {
  __shared__ float zs[kMdSpan];                  // masked metric depth of the span, 0 where invalid
  __shared__ float l0f[16 * kMdC32];
  __shared__ double l0d[2 * 32 * kMdC64];
  __shared__ int scnt[4];
  const int s = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
  const MdGeom g = md_geom(H, W);
  // several frames in one launch: mask s belongs to frame[s], which selects its depth map and camera matrix
  const int fr = frame ? frame[s] : 0;
  depth += (size_t)fr * g.n;
  K += (size_t)fr * 9;
  // the camera matrix is read on the device (3x3 row-major float64, the reference's dtype): no host copy, no cache
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const int e0 = sp * kMdSpan;
  const float4 *m4 = reinterpret_cast<const float4 *>(masks + (size_t)(msel ? msel[s] : s) * g.n + e0);   // n % 4 == 0 (launcher)
  const float4 *d4 = reinterpret_cast<const float4 *>(depth + e0);
  int cnt = 0;
  for (int i = tid; i < kMdSpan / 4; i += 256) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 + i * 4 < g.n) {
      const float4 m = m4[i], d = d4[i];
      z = make_float4(md_z(m.x, d.x, depth_scale), md_z(m.y, d.y, depth_scale), md_z(m.z, d.z, depth_scale),
                      md_z(m.w, d.w, depth_scale));
    }
    z.x = z.x > 0.f ? z.x : 0.f; z.y = z.y > 0.f ? z.y : 0.f; z.z = z.z > 0.f ? z.z : 0.f; z.w = z.w > 0.f ? z.w : 0.f;
    cnt += (z.x > 0.f) + (z.y > 0.f) + (z.z > 0.f) + (z.w > 0.f);
    reinterpret_cast<float4 *>(zs)[i] = z;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if ((tid & 63) == 0) scnt[tid >> 6] = cnt;
  __syncthreads();
  // level 0, float32: (block, column) pairs, 16 rows in row order from zero
  {
    const int col = tid & 31;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int blk = (tid >> 5) + 8 * r;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) a = __fadd_rn(a, zs[(blk * 16 + j) * kMdC32 + col]);
      l0f[blk * kMdC32 + col] = a;
    }
  }
  // level 0, float64: X = (u - cx) * Z / fx, Y = (v - cy) * Z / fy; an invalid pixel contributes (+-)0
  {
    const int col = tid & 15;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int blk = (tid >> 4) + 16 * r;
      double ax = 0.0, ay = 0.0;
#pragma unroll 4
      for (int j = 0; j < 16; ++j) {
        const int o = (blk * 16 + j) * kMdC64 + col;
        const float z = zs[o];
        const int pix = e0 + o;
        if (z > 0.f) {
          const int v = pix / W, u = pix - v * W;
          ax = __dadd_rn(ax, __ddiv_rn(__dmul_rn((double)u - cx, (double)z), fx));
          ay = __dadd_rn(ay, __ddiv_rn(__dmul_rn((double)v - cy, (double)z), fy));
        }
      }
      l0d[blk * kMdC64 + col] = ax;
      l0d[(32 + blk) * kMdC64 + col] = ay;
    }
  }
  __syncthreads();
  char *wm = ws + (size_t)s * md_ws_per_mask(g);
  if (tid < kMdC32) {                               // level 1, float32: this span's complete blocks in block order
    const int nb = min(max(g.nb32 - sp * 16, 0), 16);
    float a = 0.f;
    for (int b = 0; b < nb; ++b) a = __fadd_rn(a, l0f[b * kMdC32 + tid]);
    reinterpret_cast<float *>(wm)[sp * kMdC32 + tid] = a;
  } else if (tid >= 64 && tid < 128) {              // level 1, float64: two nodes per span, X and Y
    const int t = tid - 64, q = t >> 5, node = (t >> 4) & 1, col = t & 15;
    const int nb = min(max(g.nb64 - (sp * 2 + node) * 16, 0), 16);
    double a = 0.0;
    for (int b = 0; b < nb; ++b) a = __dadd_rn(a, l0d[(q * 32 + node * 16 + b) * kMdC64 + col]);
    reinterpret_cast<double *>(wm + md_ws_f64_off(g))[((size_t)q * 2 * g.nspan + sp * 2 + node) * kMdC64 + col] = a;
  } else if (tid == 128) {
    reinterpret_cast<int *>(wm + md_ws_cnt_off(g))[sp] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
  }
}

// Upper levels of one column: level-1 sums l1[k * stride] (k < n1; the last one partial when nb % 16 != 0) -> the
// column's value after `acc[0] += acc[1]; += acc[2]; += acc[3]`, given the column's row remainder `rem`.
template <typename T>
__device__ __forceinline__ T md_upper(const T *l1, int stride, int nb, T rem) {
  const int nfull = nb / 16, n1 = (nb + 15) / 16;
  T a2 = 0, a3 = 0;
  for (int k = 0; k < nfull; ++k) {
    a2 = a2 + l1[(size_t)k * stride];
    if (((k + 1) & 15) == 0) { a3 = a3 + a2; a2 = 0; }
  }
  const T a1 = n1 > nfull ? l1[(size_t)nfull * stride] : (T)0;
  return ((rem + a1) + a2) + a3;
}

__global__ __launch_bounds__(64) void masked_depth_final_kernel(const float *__restrict__ masks, const float *__restrict__ depth,
                                                               int H, int W, float depth_scale, const double *__restrict__ K,
                                                               const int *__restrict__ frame, const int *__restrict__ msel,
                                                               const char *__restrict__ ws, float *__restrict__ out) {
  __shared__ float cz[kMdC32];
  __shared__ double cxy[2 * kMdC64];
  const int s = blockIdx.x, lane = threadIdx.x;
  const MdGeom g = md_geom(H, W);
  const int fr = frame ? frame[s] : 0;
  depth += (size_t)fr * g.n;
  K += (size_t)fr * 9;
  masks += (size_t)(msel ? msel[s] : s) * g.n;
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  const char *wm = ws + (size_t)s * md_ws_per_mask(g);
  auto zat = [&](int i) { const float z = md_z(masks[i], depth[i], depth_scale); return z > 0.f ? z : 0.f; };
  auto xat = [&](int i) { const float z = zat(i); return z > 0.f ? ((double)(i % W) - cx) * (double)z / fx : 0.0; };
  auto yat = [&](int i) { const float z = zat(i); return z > 0.f ? ((double)(i / W) - cy) * (double)z / fy : 0.0; };
  if (lane < kMdC32) {
    const int rows = (g.n / 8) / 4;
    float rem = 0.f;
    for (int r = g.nb32 * 16; r < rows; ++r) rem = rem + zat(r * kMdC32 + lane);
    cz[lane] = md_upper<float>(reinterpret_cast<const float *>(wm) + lane, kMdC32, g.nb32, rem);
  } else {
    const int q = (lane - 32) >> 4, col = lane & 15, rows = (g.n / 4) / 4;
    double rem = 0.0;
    for (int r = g.nb64 * 16; r < rows; ++r) rem = rem + (q ? yat(r * kMdC64 + col) : xat(r * kMdC64 + col));
    cxy[q * kMdC64 + col] = md_upper<double>(
        reinterpret_cast<const double *>(wm + md_ws_f64_off(g)) + (size_t)q * 2 * g.nspan * kMdC64 + col, kMdC64, g.nb64, rem);
  }
  __syncthreads();
  // vectors 1..3 onto vector 0 (after the tail vectors of the row remainder), then the lanes in order onto the scalar tail
  __shared__ double res[4];                                 // X, Y, Z sums and the float32 count
  if (lane == 0) {
    const int *pc = reinterpret_cast<const int *>(wm + md_ws_cnt_off(g));
    int c = 0;
    for (int i = 0; i < g.nspan; ++i) c += pc[i];
    res[3] = (double)((float)c + 1e-8f);                    // count_nonzero + 1e-8 is float32 in the reference
    const int vec = g.n / 8;
    float ps[8];
    for (int l = 0; l < 8; ++l) {
      float a = cz[l];
      for (int r = (vec / 4) * 4; r < vec; ++r) a = a + zat(r * 8 + l);
      ps[l] = ((a + cz[8 + l]) + cz[16 + l]) + cz[24 + l];
    }
    float f = 0.f;
    for (int i = vec * 8; i < g.n; ++i) f = f + zat(i);
    for (int l = 0; l < 8; ++l) f = f + ps[l];
    res[2] = (double)f;
  }
  if (lane == 1 || lane == 2) {
    const int q = lane - 1, vec = g.n / 4;
    const double *c = cxy + q * kMdC64;
    double ps[4];
    for (int l = 0; l < 4; ++l) {
      double a = c[l];
      for (int r = (vec / 4) * 4; r < vec; ++r) a = a + (q ? yat(r * 4 + l) : xat(r * 4 + l));
      ps[l] = ((a + c[4 + l]) + c[8 + l]) + c[12 + l];
    }
    double f = 0.0;
    for (int i = vec * 4; i < g.n; ++i) f = f + (q ? yat(i) : xat(i));
    for (int l = 0; l < 4; ++l) f = f + ps[l];
    res[q] = f;
  }
  __syncthreads();
  if (lane == 0) {
    const float cnt = (float)res[3];
    out[s * 3 + 0] = (float)(res[0] / (double)cnt);
    out[s * 3 + 1] = (float)(res[1] / (double)cnt);
    out[s * 3 + 2] = (float)res[2] / cnt;
  }
}

// Rotate the object's model points by the best template pose, translate, project with K, truncate to
// pixels, clamp, and reduce the per-proposal bounding box (detector.py:209-232 + :316-318).
__global__ __launch_bounds__(256) void project_bbox_kernel(const float *__restrict__ pointcloud, const float *__restrict__ poses,
                                                          const int *__restrict__ obj, const int *__restrict__ tmpl,
                                                          const float *__restrict__ trans, const float *__restrict__ K,
                                                          const int *__restrict__ frame, int N, int H, int W,
                                                          int *__restrict__ uv, int *__restrict__ bbox) {
  __shared__ int red[4][4];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (frame) K += (size_t)frame[s] * 9;                      // several frames in one launch: one camera matrix per frame
  const float *R = poses + (size_t)tmpl[s] * 16;            // 4x4 row-major, rotation in [0:3,0:3]
  const float *pc = pointcloud + (size_t)obj[s] * N * 3;
  const float tx = trans[s * 3], ty = trans[s * 3 + 1], tz = trans[s * 3 + 2];
  int mnu = 1 << 30, mnv = 1 << 30, mxu = -(1 << 30), mxv = -(1 << 30);
  for (int i = tid; i < N; i += 256) {
    const float x = pc[i * 3], y = pc[i * 3 + 1], z = pc[i * 3 + 2];
    // Operation order of the reference's statements on its pinned (CPU) run, spelled out so that the float -> int
    // truncation sees the same bits: each 3-term product is a k-ascending fma chain (bmm, detector.py:218-221),
    // the translation is a separate rounded add (:222), then the second product (:226), an IEEE division (:227), .to(int).
    const float px = __fadd_rn(__fmaf_rn(R[2], z, __fmaf_rn(R[1], y, __fmul_rn(R[0], x))), tx);
    const float py = __fadd_rn(__fmaf_rn(R[6], z, __fmaf_rn(R[5], y, __fmul_rn(R[4], x))), ty);
    const float pz = __fadd_rn(__fmaf_rn(R[10], z, __fmaf_rn(R[9], y, __fmul_rn(R[8], x))), tz);
    const float hx = __fmaf_rn(K[2], pz, __fmaf_rn(K[1], py, __fmul_rn(K[0], px)));
    const float hy = __fmaf_rn(K[5], pz, __fmaf_rn(K[4], py, __fmul_rn(K[3], px)));
    const float hz = __fmaf_rn(K[8], pz, __fmaf_rn(K[7], py, __fmul_rn(K[6], px)));
    int u = (int)__fdiv_rn(hx, hz), v = (int)__fdiv_rn(hy, hz);   // .to(torch.int): truncation toward zero
    u = min(max(u, 0), W - 1);
    v = min(max(v, 0), H - 1);
    uv[((size_t)s * N + i) * 2] = u;
    uv[((size_t)s * N + i) * 2 + 1] = v;
    mnu = min(mnu, u); mxu = max(mxu, u); mnv = min(mnv, v); mxv = max(mxv, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mnu = min(mnu, __shfl_xor(mnu, off)); mnv = min(mnv, __shfl_xor(mnv, off));
    mxu = max(mxu, __shfl_xor(mxu, off)); mxv = max(mxv, __shfl_xor(mxv, off));
  }
  if ((tid & 63) == 0) { red[tid >> 6][0] = mnu; red[tid >> 6][1] = mnv; red[tid >> 6][2] = mxu; red[tid >> 6][3] = mxv; }
  __syncthreads();
  if (tid == 0) {
    bbox[s * 4 + 0] = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    bbox[s * 4 + 1] = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
    bbox[s * 4 + 2] = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
    bbox[s * 4 + 3] = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_pairwise_cosine_f32(const float *query, const float *ref, int P, int R, int C, float *out,
                                       void *stream) {
  if (P < 0 || R < 0 || C <= 0 || (C % 16) != 0) return S6D_EINVAL;
  if ((size_t)P * R == 0) return S6D_OK;
  if (!query || !ref || !out) return S6D_EINVAL;
  const int tiles = ((P + 15) / 16) * ((R + 15) / 16);
  hipLaunchKernelGGL(pairwise_cosine_kernel, dim3((tiles + 3) / 4), dim3(256), 0, as_stream(stream), query, ref, P, R, C, out);
  return launch_status();
}

extern "C" int s6d_semantic_select_f32(const float *scores, int P, int O, int T, int topk, float *best_score,
                                       int32_t *best_obj, int32_t *best_tmpl, void *stream) {
  if (P < 0 || O <= 0 || T <= 0 || topk <= 0) return S6D_EINVAL;
  if (T > 256) return S6D_EUNSUPPORTED;
  if (P == 0) return S6D_OK;
  if (!scores || !best_score || !best_obj || !best_tmpl) return S6D_EINVAL;
  hipLaunchKernelGGL(semantic_select_kernel, dim3((P + 3) / 4), dim3(256), 0, as_stream(stream), scores, P, O, T, topk,
                     best_score, best_obj, best_tmpl);
  return launch_status();
}

extern "C" int s6d_patch_scores_sel_f32(const float *query, const int32_t *qsel, const float *refstore, const int32_t *obj,
                                        const int32_t *tmpl, int S, int N1, int N2, int C, int T, float thred, float *workspace,
                                        float *appe, float *ratio, void *stream);

extern "C" int s6d_patch_scores_f32(const float *query, const float *refstore, const int32_t *obj, const int32_t *tmpl,
                                    int S, int N1, int N2, int C, int T, float thred, float *workspace, float *appe,
                                    float *ratio, void *stream) {
  return s6d_patch_scores_sel_f32(query, nullptr, refstore, obj, tmpl, S, N1, N2, C, T, thred, workspace, appe, ratio, stream);
}

extern "C" int s6d_patch_scores_sel_f32(const float *query, const int32_t *qsel, const float *refstore, const int32_t *obj,
                                        const int32_t *tmpl, int S, int N1, int N2, int C, int T, float thred, float *workspace,
                                        float *appe, float *ratio, void *stream) {
  if (S < 0 || N1 <= 0 || N2 <= 0 || C <= 0 || (C % 32) != 0 || T <= 0) return S6D_EINVAL;
  if (N2 > 16 * kMaxColTiles) return S6D_EUNSUPPORTED;
  if (S == 0) return S6D_OK;
  if (!query || !refstore || !obj || !tmpl || !workspace || !appe || !ratio) return S6D_EINVAL;
  const int nrb = (N1 + 255) / 256, nslot = nrb * 8;
  // workspace: [S*nslot] row sums | [S*nslot*N2] column maxima | [S*nslot] non-zero row counts (int)
  float *part_rowsum = workspace;
  float *part_colmax = workspace + (size_t)S * nslot;
  int *part_nnz = reinterpret_cast<int *>(part_colmax + (size_t)S * nslot * N2);
  hipStream_t st = as_stream(stream);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&patch_scores_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kPsLdsBytes);
  hipLaunchKernelGGL(patch_scores_kernel, dim3(nrb, S), dim3(kPsThreads), kPsLdsBytes, st, query, refstore, obj, tmpl, qsel,
                     N1, N2, C, T, part_rowsum, part_colmax, part_nnz);
  int rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(patch_finalize_kernel, dim3(S), dim3(64), 0, st, part_rowsum, part_colmax, part_nnz, S, nslot, N1, N2,
                     thred, appe, ratio);
  return launch_status();
}

extern "C" long s6d_patch_scores_workspace_floats(int S, int N1, int N2) {
  const long nslot = ((N1 + 255) / 256) * 8;
  return (long)S * nslot * (2 + N2);
}

extern "C" long s6d_masked_depth_mean_workspace_bytes(int S, int H, int W) {
  if (S < 0 || H <= 0 || W <= 0) return 0;
  return (long)((size_t)S * md_ws_per_mask(md_geom(H, W)));
}

extern "C" int s6d_masked_depth_mean_sel_f32(const float *masks, const int32_t *msel, const float *depth, const int32_t *frame, int S,
                                             int H, int W, float depth_scale, const double *K, void *workspace, float *out,
                                             void *stream);

extern "C" int s6d_masked_depth_mean_f32(const float *masks, const float *depth, int S, int H, int W, float depth_scale,
                                         const double *K, void *workspace, float *out, void *stream) {
  return s6d_masked_depth_mean_sel_f32(masks, nullptr, depth, nullptr, S, H, W, depth_scale, K, workspace, out, stream);
}

extern "C" int s6d_masked_depth_mean_frames_f32(const float *masks, const float *depth, const int32_t *frame, int S, int H, int W,
                                                float depth_scale, const double *K, void *workspace, float *out, void *stream) {
  return s6d_masked_depth_mean_sel_f32(masks, nullptr, depth, frame, S, H, W, depth_scale, K, workspace, out, stream);
}

extern "C" int s6d_masked_depth_mean_sel_f32(const float *masks, const int32_t *msel, const float *depth, const int32_t *frame, int S,
                                             int H, int W, float depth_scale, const double *K, void *workspace, float *out,
                                             void *stream) {
  if (S < 0 || H <= 0 || W <= 0) return S6D_EINVAL;
  // float4 loads of the maps; one vector of the float32 reduction at least; 16 rows per cascade level (ATen switches to
  // 32 from 2^20 rows of the float64 reduction on: 16.7 M pixels)
  if ((W % 4) != 0 || (long)H * W < 8 || (long)H * W >= (1l << 24)) return S6D_EUNSUPPORTED;
  if (S == 0) return S6D_OK;
  if (!masks || !depth || !out || !workspace || !K) return S6D_EINVAL;
  const MdGeom g = md_geom(H, W);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(masked_depth_l1_kernel, dim3(g.nspan, S), dim3(256), 0, st, masks, depth, H, W, depth_scale, K,
                     frame, msel, (char *)workspace);
  int rc = launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(masked_depth_final_kernel, dim3(S), dim3(64), 0, st, masks, depth, H, W, depth_scale, K, frame,
                     msel, (const char *)workspace, out);
  return launch_status();
}

extern "C" int s6d_project_bbox_frames_f32(const float *pointcloud, const float *poses, const int32_t *obj, const int32_t *tmpl,
                                           const float *trans, const float *K, const int32_t *frame, int S, int N, int H, int W,
                                           int32_t *uv, int32_t *bbox, void *stream);

extern "C" int s6d_project_bbox_f32(const float *pointcloud, const float *poses, const int32_t *obj, const int32_t *tmpl,
                                    const float *trans, const float *K, int S, int N, int H, int W, int32_t *uv,
                                    int32_t *bbox, void *stream) {
  return s6d_project_bbox_frames_f32(pointcloud, poses, obj, tmpl, trans, K, nullptr, S, N, H, W, uv, bbox, stream);
}

extern "C" int s6d_project_bbox_frames_f32(const float *pointcloud, const float *poses, const int32_t *obj, const int32_t *tmpl,
                                           const float *trans, const float *K, const int32_t *frame, int S, int N, int H, int W,
                                           int32_t *uv, int32_t *bbox, void *stream) {
  if (S < 0 || N <= 0 || H <= 0 || W <= 0) return S6D_EINVAL;
  if (S == 0) return S6D_OK;
  if (!pointcloud || !poses || !obj || !tmpl || !trans || !K || !uv || !bbox) return S6D_EINVAL;
  hipLaunchKernelGGL(project_bbox_kernel, dim3(S), dim3(256), 0, as_stream(stream), pointcloud, poses, obj, tmpl, trans, K, frame,
                     N, H, W, uv, bbox);
  return launch_status();
}
