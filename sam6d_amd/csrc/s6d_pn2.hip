// PointNet++ ops of the PEM hot path, written for gfx950 (MI355X).
//
// Replaces the reference's CUDA-only extension
// (Pose_Estimation_Model/model/pointnet2/_ext_src/src/{sampling,ball_query,group_points}_gpu.cu).
// Results are integer-exact with the reference's selection rules; the structure is not
// the reference's (one-thread-per-centre scans, 512-thread shared-memory trees) but
// wave64-native: register-resident clouds, ballot/prefix compaction, 64-bit key reductions.
//
// Floating point: distances are the fma chain nvcc emits for the reference expression
// (see oracle/pn2_oracle.c header); contraction of anything else is disabled here.
#pragma clang fp contract(off)

#include "s6d_common.h"

namespace s6d {

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
}

// Selection key: larger distance wins.  Ties follow the reference's reduction exactly: a thread
// keeps its FIRST maximum (strict '>', sampling_gpu.cu:113-114) and the shared-memory tree
// halves the stride from block_size/2 down to 1 with the lower slot winning (__update,
// :64-70), so two tied slots meet at the stride of their lowest differing bit and the slot
// whose bit is 0 survives: the winner is the smallest BIT-REVERSED slot id, then the lowest k
// within that thread.  ref_bs = opt_n_threads(N) (cuda_utils.h:18-23).
__device__ __forceinline__ unsigned long long fps_key(float d, int k, int ref_bs_log2) {
  const unsigned tid = (unsigned)k & ((1u << ref_bs_log2) - 1u);
  const unsigned rev = ref_bs_log2 ? (__brev(tid) >> (32 - ref_bs_log2)) : 0u;
  const unsigned q = (unsigned)k >> ref_bs_log2;
  const unsigned tie = (rev << 22) | q;  // q < 2^22  <=>  N < 2^31 for ref_bs = 512
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xffffffffu - tie);
}
__device__ __forceinline__ int fps_key_index(unsigned long long key, int ref_bs_log2) {
  const unsigned tie = 0xffffffffu - (unsigned)(key & 0xffffffffull);
  const unsigned rev = tie >> 22;
  const unsigned tid = ref_bs_log2 ? (__brev(rev) >> (32 - ref_bs_log2)) : 0u;
  return (int)(((tie & 0x3fffffu) << ref_bs_log2) | tid);
}

// One workgroup per cloud; the cloud and its running min-distances live in registers
// (PPT points per thread), a copy of the coordinates in LDS serves the "last selected
// point" broadcast.  One barrier per selected point.
template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(const float *__restrict__ xyz, int N, int M,
                                                         int ref_bs_log2, int32_t *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *spts = reinterpret_cast<float *>(smem);                               // N*3
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem + ((N * 3 * 4 + 15) / 16) * 16);
  constexpr int NW = THREADS / kWave;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *p = xyz + (size_t)b * N * 3;
  int32_t *out = idx + (size_t)b * M;

  for (int i = tid; i < N * 3; i += THREADS) spts[i] = p[i];
  __syncthreads();
  float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    const int k = tid + r * THREADS;
    const bool ok = k < N;
    px[r] = ok ? spts[k * 3 + 0] : 0.f;
    py[r] = ok ? spts[k * 3 + 1] : 0.f;
    pz[r] = ok ? spts[k * 3 + 2] : 0.f;
    td[r] = 1e10f;
  }
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < M; ++j) {
    const float x1 = spts[old * 3 + 0], y1 = spts[old * 3 + 1], z1 = spts[old * 3 + 2];
    unsigned long long best = 0ull;  // below every real key (real keys have non-zero low word)
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int k = tid + r * THREADS;
      const float d = sqdist3(px[r], py[r], pz[r], x1, y1, z1);
      const float d2 = fminf(d, td[r]);
      td[r] = d2;
      const unsigned long long key = fps_key(d2, k, ref_bs_log2);
      best = (k < N && key > best) ? key : best;
    }
    best = wave_max_u64(best);
    unsigned long long *sl = slots + (j & 1) * NW;
    if (lane_id() == 0) sl[tid / kWave] = best;
    __syncthreads();
    unsigned long long m = sl[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = sl[w] > m ? sl[w] : m;
    old = fps_key_index(m, ref_bs_log2);
    if (tid == 0) out[j] = old;
  }
}

// Any N: coordinates and running distances stay in global memory / L2 (template onboarding,
// 210000 -> 2048 points, Pose_Estimation_Model/model/feature_extraction.py:170-181).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fps_mem_kernel(const float *__restrict__ xyz, int N, int M,
                                                         int ref_bs_log2, float *__restrict__ tmp,
                                                         int32_t *__restrict__ idx) {
  __shared__ unsigned long long slots[2][THREADS / kWave];
  constexpr int NW = THREADS / kWave;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *p = xyz + (size_t)b * N * 3;
  float *td = tmp + (size_t)b * N;
  int32_t *out = idx + (size_t)b * M;
  for (int k = tid; k < N; k += THREADS) td[k] = 1e10f;
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < M; ++j) {
    const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
    unsigned long long best = 0ull;
    for (int k = tid; k < N; k += THREADS) {  // a thread only ever touches its own td[k]
      const float d = sqdist3(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], x1, y1, z1);
      const float d2 = fminf(d, td[k]);
      td[k] = d2;
      const unsigned long long key = fps_key(d2, k, ref_bs_log2);
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if (lane_id() == 0) slots[j & 1][tid / kWave] = best;
    __syncthreads();
    unsigned long long m = slots[j & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = slots[j & 1][w] > m ? slots[j & 1][w] : m;
    old = fps_key_index(m, ref_bs_log2);
    if (tid == 0) out[j] = old;
  }
}

__global__ void gather_points_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                     int C, int N, int M, size_t total, float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % M);
    const size_t bc = i / M;
    const size_t b = bc / C;
    out[i] = points[bc * N + idx[b * M + j]];
  }
}

// src (B,N,C) -> out (B,M,C): one 16-byte chunk per thread, rows are contiguous.
__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx, int N, int C,
                                   int M, size_t total_vec, int vec, float *__restrict__ out) {
  const int cv = C / vec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    const size_t bm = i / cv;
    const size_t b = bm / M;
    const size_t row = b * N + idx[bm];
    if (vec == 4) {
      reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(src)[row * cv + c];
    } else {
      out[i] = src[row * cv + c];
    }
  }
}

// One wavefront per query centre: 64 candidates are tested per step, hits are compacted in
// scan order with a ballot + prefix popcount, so the output equals the sequential first-
// nsample scan of the reference (ball_query_gpu.cu:31-47).  The cloud is staged in LDS once
// per workgroup and shared by its waves.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void ball_query_kernel(const float *__restrict__ new_xyz,
                                                            const float *__restrict__ xyz, int N, int M,
                                                            float radius2, int nsample, int centres_per_block,
                                                            int32_t *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *spts = reinterpret_cast<float *>(smem);
  const int b = blockIdx.y;
  const float *p = xyz + (size_t)b * N * 3;
  for (int i = threadIdx.x; i < N * 3; i += THREADS) spts[i] = p[i];
  __syncthreads();
  const int wave = threadIdx.x / kWave, lane = lane_id();
  constexpr int NW = THREADS / kWave;
  const int j0 = blockIdx.x * centres_per_block;
  for (int jj = wave; jj < centres_per_block; jj += NW) {
    const int j = j0 + jj;
    if (j >= M) break;
    const float *q = new_xyz + ((size_t)b * M + j) * 3;
    const float cx = q[0], cy = q[1], cz = q[2];
    int32_t *o = idx + ((size_t)b * M + j) * nsample;
    int cnt = 0, first = 0;
    for (int k0 = 0; k0 < N && cnt < nsample; k0 += kWave) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < N) hit = sqdist3(cx, cy, cz, spts[k * 3 + 0], spts[k * 3 + 1], spts[k * 3 + 2]) < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        if (cnt == 0) first = k0 + __ffsll((long long)mask) - 1;
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < nsample) o[pos] = k;
        cnt += __popcll(mask);
      }
    }
    cnt = cnt < nsample ? cnt : nsample;
    // first-hit fill (or zeros when nothing is in range: the reference output is torch::zeros)
    for (int l = cnt + lane; l < nsample; l += kWave) o[l] = first;
  }
}

__global__ void group_points_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx, int C,
                                    int N, int M, int S, size_t total, float *__restrict__ out) {
  const size_t ms = (size_t)M * S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t jk = i % ms;
    const size_t bc = i / ms;
    const size_t b = bc / C;
    out[i] = points[bc * N + idx[b * ms + jk]];
  }
}

static int ilog2_floor(int v) {
  int l = 0;
  while ((2 << l) <= v) ++l;
  return l;
}
static int grid_for(size_t total, int threads) {
  size_t g = (total + threads - 1) / threads;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_fps_f32(const float *xyz, int B, int N, int M, float *tmp, int32_t *idx, void *stream) {
  if (B < 0 || N <= 0 || M < 0 || M > N) return S6D_EINVAL;
  if (B == 0 || M == 0) return S6D_OK;
  if (!xyz || !idx) return S6D_EINVAL;
  int bs_log2 = ilog2_floor(N);  // opt_n_threads(N) = 2^min(floor(log2 N), 9)  (cuda_utils.h:18-23)
  if (bs_log2 > 9) bs_log2 = 9;
  hipStream_t st = as_stream(stream);
  if (N <= 2048) {
    const size_t lds = ((size_t)N * 12 + 15) / 16 * 16 + 2 * 4 * sizeof(unsigned long long);
    hipLaunchKernelGGL((fps_reg_kernel<256, 8>), dim3(B), dim3(256), lds, st, xyz, N, M, bs_log2, idx);
  } else if (N <= 4096) {
    const size_t lds = ((size_t)N * 12 + 15) / 16 * 16 + 2 * 8 * sizeof(unsigned long long);
    hipLaunchKernelGGL((fps_reg_kernel<512, 8>), dim3(B), dim3(512), lds, st, xyz, N, M, bs_log2, idx);
  } else {
    if (!tmp) return S6D_EINVAL;
    hipLaunchKernelGGL((fps_mem_kernel<1024>), dim3(B), dim3(1024), 0, st, xyz, N, M, bs_log2, tmp, idx);
  }
  return launch_status();
}

extern "C" int s6d_gather_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int M,
                                     float *out, void *stream) {
  if (B < 0 || C < 0 || N <= 0 || M < 0) return S6D_EINVAL;
  const size_t total = (size_t)B * C * M;
  if (total == 0) return S6D_OK;
  if (!points || !idx || !out) return S6D_EINVAL;
  hipLaunchKernelGGL(gather_points_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), points, idx,
                     C, N, M, total, out);
  return launch_status();
}

extern "C" int s6d_gather_rows_f32(const float *src, const int32_t *idx, int B, int N, int C, int M, float *out,
                                   void *stream) {
  if (B < 0 || C <= 0 || N <= 0 || M < 0) return S6D_EINVAL;
  if ((size_t)B * M == 0) return S6D_OK;
  if (!src || !idx || !out) return S6D_EINVAL;
  const int vec = (C % 4 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0)) ? 4 : 1;
  const size_t total = (size_t)B * M * (C / vec);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), src, idx, N, C,
                     M, total, vec, out);
  return launch_status();
}

extern "C" int s6d_ball_query_f32(const float *new_xyz, const float *xyz, int B, int N, int M, float radius,
                                  int nsample, int32_t *idx, void *stream) {
  if (B < 0 || N <= 0 || M < 0 || nsample <= 0) return S6D_EINVAL;
  if ((size_t)B * M == 0) return S6D_OK;
  if (!new_xyz || !xyz || !idx) return S6D_EINVAL;
  if ((size_t)N * 12 > 64 * 1024) return S6D_EUNSUPPORTED;  // cloud is staged in LDS (N <= 5461)
  const int cpb = 32;
  dim3 grid((M + cpb - 1) / cpb, B);
  hipLaunchKernelGGL((ball_query_kernel<256>), grid, dim3(256), (size_t)N * 12, as_stream(stream), new_xyz, xyz, N,
                     M, radius * radius, nsample, cpb, idx);
  return launch_status();
}

extern "C" int s6d_group_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int M, int S,
                                    float *out, void *stream) {
  if (B < 0 || C < 0 || N <= 0 || M < 0 || S < 0) return S6D_EINVAL;
  const size_t total = (size_t)B * C * M * S;
  if (total == 0) return S6D_OK;
  if (!points || !idx || !out) return S6D_EINVAL;
  hipLaunchKernelGGL(group_points_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), points, idx, C,
                     N, M, S, total, out);
  return launch_status();
}
