// Kernels of the PEM per-detection pre-processing (host side: sam6d_amd/pem/preprocess.py).
//
//   sample_indices   the point sampler of get_test_data / get_instance (Pose_Estimation_Model/run_inference_custom.py:224-229,
//                    provider/bop_test_dataset.py:140-145) in its DEFINED form (preprocess.py header: injected uniforms
//                    instead of numpy's global RNG): per detection with n candidate points and keys u_0 .. u_{n-1},
//                      n <= n_sample : idx_i = floor(u_i * n) for i < n_sample            (with replacement)
//                      n >  n_sample : the positions of the n_sample smallest (u, position) pairs, ascending
//                                      (without replacement; == the first n_sample entries of a stable argsort).
//                    The library version is one top-k over a (P, L) table of 64-bit composite keys -- a full sort of up to
//                    3e5 entries per detection.  Here one workgroup per detection reads its keys twice: a 4096-bin histogram of
//                    the keys' leading bits finds the bin in which the n_sample-th smallest key lies, the (at most 4096) keys
//                    up to that bin are collected in LDS and sorted there (bitonic, 64-bit composite key << 32 | position).
//                    Exact, deterministic, independent of the order of the atomics.  Keys must be non-negative floats (their
//                    bit patterns then order like their values); if more than 4096 keys share the leading bits up to the
//                    threshold bin (heavily duplicated keys, never uniform ones) the detection is flagged in `overflow` and the
//                    caller uses the library path for it.
//   compact_cloud    "mask -> choose -> cloud" of get_test_data (:209-213): the masked pixels of a detection's square crop in
//                    row-major crop order (``mask[y1:y2, x1:x2].flatten().nonzero()``) with their back-projected points
//                    (utils/data_utils.py:92-110: x = (u - cx) z / fx, y = (v - cy) z / fy in float32, that operation order).
//                    The library version builds ONE list for the frame with nonzero + five boolean-mask indexings (a host round
//                    trip each).  Here one workgroup per detection walks its crop in chunks of 1024 pixels and compacts in
//                    order (wave ballot ranks + a 16-entry wave prefix), writing to the detection's fixed-capacity slot.
//   radius_filter    ``flag = norm(cloud - center) < radius * 1.2`` (:214-221) and the second in-order compaction, in place.
#include "s6d_common.h"

namespace s6d {

#pragma clang fp contract(off)   // the reference's float32 expressions, operation by operation (no fused multiply-adds)

constexpr int kCmpThreads = 1024;

// in-order compaction step of one 1024-element chunk: returns this lane's output position (or -1) and advances *base
__device__ __forceinline__ long block_rank(bool keep, long *base, unsigned *wave_tot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long bal = __ballot(keep);
  const int rank = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = (unsigned)__popcll(bal);
  __syncthreads();
  long off = *base;
  unsigned tot = 0;
  for (int w = 0; w < kCmpThreads / 64; ++w) {
    if (w < wave) off += wave_tot[w];
    tot += wave_tot[w];
  }
  __syncthreads();                                      // everyone has read base / wave_tot before they change
  if (tid == 0) *base += tot;
  return keep ? off + rank : -1;
}

// mask (P,H,W) u8 (non-zero = set), depth (H,W) f32 -> m (P,H,W) u8 = mask AND depth > 0, cnt (P) i64, ok (P) u8 = cnt > min_points,
// box (P,4) i64: get_bbox (utils/data_utils.py:126-160) of m -- of the whole frame for a detection that is not ok, like the
// library path's dummy mask.  One workgroup per detection, one pass over its mask.
__global__ __launch_bounds__(kCmpThreads) void mask_boxes_kernel(const unsigned char *__restrict__ mask,
                                                                const float *__restrict__ depth, int H, int W, long min_points,
                                                                unsigned char *__restrict__ m, long *__restrict__ cnt_out,
                                                                unsigned char *__restrict__ ok_out, long *__restrict__ box) {
  __shared__ int s_cnt[kCmpThreads / 64], s_y0[kCmpThreads / 64], s_y1[kCmpThreads / 64], s_x0[kCmpThreads / 64],
      s_x1[kCmpThreads / 64];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t off = (size_t)p * H * W;
  int cnt = 0, y0 = H, y1 = -1, x0 = W, x1 = -1;
  for (int i = tid; i < H * W; i += kCmpThreads) {
    const bool v = mask[off + i] != 0 && depth[i] > 0.f;
    m[off + i] = v ? 1 : 0;
    if (v) {
      const int y = i / W, x = i - y * W;
      ++cnt;
      y0 = min(y0, y);
      y1 = max(y1, y);
      x0 = min(x0, x);
      x1 = max(x1, x);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o);
    y0 = min(y0, __shfl_xor(y0, o));
    y1 = max(y1, __shfl_xor(y1, o));
    x0 = min(x0, __shfl_xor(x0, o));
    x1 = max(x1, __shfl_xor(x1, o));
  }
  if (lane == 0) {
    s_cnt[wave] = cnt;
    s_y0[wave] = y0;
    s_y1[wave] = y1;
    s_x0[wave] = x0;
    s_x1[wave] = x1;
  }
  __syncthreads();
  if (tid == 0) {
    long c = 0;
    int ry0 = H, ry1 = -1, rx0 = W, rx1 = -1;
    for (int w = 0; w < kCmpThreads / 64; ++w) {
      c += s_cnt[w];
      ry0 = min(ry0, s_y0[w]);
      ry1 = max(ry1, s_y1[w]);
      rx0 = min(rx0, s_x0[w]);
      rx1 = max(rx1, s_x1[w]);
    }
    const bool ok = c > min_points;
    long rmin = ok ? ry0 : 0, rmax = ok ? ry1 + 1 : H, cmin = ok ? rx0 : 0, cmax = ok ? rx1 + 1 : W;
    const long rb = rmax - rmin, cb = cmax - cmin, lim = H < W ? H : W;
    long b = rb > cb ? rb : cb;
    b = b < lim ? b : lim;
    const long cy = (rmin + rmax) / 2, cx = (cmin + cmax) / 2, half = b / 2;
    rmin = cy - half;
    rmax = cy + half;
    cmin = cx - half;
    cmax = cx + half;
    if (rmin < 0) {
      rmax -= rmin;
      rmin = 0;
    }
    if (cmin < 0) {
      cmax -= cmin;
      cmin = 0;
    }
    if (rmax > H) {
      rmin -= rmax - H;
      rmax = H;
    }
    if (cmax > W) {
      cmin -= cmax - W;
      cmax = W;
    }
    cnt_out[p] = c;
    ok_out[p] = ok ? 1 : 0;
    box[p * 4 + 0] = rmin;
    box[p * 4 + 1] = rmax;
    box[p * 4 + 2] = cmin;
    box[p * 4 + 3] = cmax;
  }
}

// m (P,H,W) u8 = mask AND depth > 0; box (P,4) i64 [y1,y2,x1,x2]; ok (P) u8 -> choose (P,cap) i32, cloud (P,cap,3) f32, n (P) i64
__global__ __launch_bounds__(kCmpThreads) void compact_cloud_kernel(const unsigned char *__restrict__ m,
                                                                   const float *__restrict__ depth,
                                                                   const long *__restrict__ box,
                                                                   const unsigned char *__restrict__ ok, int H, int W,
                                                                   float fx, float fy, float cx, float cy, long cap,
                                                                   int *__restrict__ choose, float *__restrict__ cloud,
                                                                   long *__restrict__ n_out) {
  __shared__ unsigned wave_tot[kCmpThreads / 64];
  __shared__ long base;
  const int p = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) base = 0;
  __syncthreads();
  if (ok[p]) {
    const long y1 = box[p * 4 + 0], y2 = box[p * 4 + 1], x1 = box[p * 4 + 2], x2 = box[p * 4 + 3];
    const int bw = (int)(x2 - x1), area = (int)(y2 - y1) * bw;    // a crop is at most min(H,W)^2 pixels: 32-bit index arithmetic
    const unsigned char *mp = m + (size_t)p * H * W;
    for (int c0 = 0; c0 < area; c0 += kCmpThreads) {
      const int j = c0 + tid;
      bool keep = false;
      int y = 0, x = 0;
      if (j < area) {
        const int r = j / bw;
        y = (int)y1 + r;
        x = (int)x1 + (j - r * bw);
        keep = mp[y * W + x] != 0;
      }
      const long pos = block_rank(keep, &base, wave_tot);
      if (keep) {
        const float z = depth[y * W + x];
        choose[(size_t)p * cap + pos] = j;
        float *c = cloud + ((size_t)p * cap + pos) * 3;
        c[0] = ((float)x - cx) * z / fx;
        c[1] = ((float)y - cy) * z / fy;
        c[2] = z;
      }
    }
  }
  __syncthreads();
  if (tid == 0) n_out[p] = base;
}

// in place: keep the points within lim[p] of center[p], in order.  choose (P,cap) i32, cloud (P,cap,3) f32, n (P) i64 in/out
__global__ __launch_bounds__(kCmpThreads) void radius_filter_kernel(const float *__restrict__ center,
                                                                   const double *__restrict__ lim, long cap,
                                                                   int *__restrict__ choose, float *__restrict__ cloud,
                                                                   long *__restrict__ n_io) {
  __shared__ unsigned wave_tot[kCmpThreads / 64];
  __shared__ long base;
  const int p = blockIdx.x, tid = threadIdx.x;
  const long n = n_io[p];
  if (tid == 0) base = 0;
  __syncthreads();
  const float c0 = center[p * 3 + 0], c1 = center[p * 3 + 1], c2 = center[p * 3 + 2];
  const double limit = lim[p];
  int *ch = choose + (size_t)p * cap;
  float *cl = cloud + (size_t)p * cap * 3;
  for (long s0 = 0; s0 < n; s0 += kCmpThreads) {
    const long i = s0 + tid;
    bool keep = false;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    int cj = 0;
    if (i < n) {
      vx = cl[i * 3 + 0];
      vy = cl[i * 3 + 1];
      vz = cl[i * 3 + 2];
      cj = ch[i];
      const float dx = vx - c0, dy = vy - c1, dz = vz - c2;
      const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
      keep = (double)d < limit;
    }
    const long pos = block_rank(keep, &base, wave_tot);   // its barriers also order this chunk's reads before the writes below
    if (keep) {                                           // pos <= i: never overwrites an element that is still to be read
      ch[pos] = cj;
      cl[pos * 3 + 0] = vx;
      cl[pos * 3 + 1] = vy;
      cl[pos * 3 + 2] = vz;
    }
  }
  __syncthreads();
  if (tid == 0) n_io[p] = base;
}

constexpr int kSelThreads = 1024;
constexpr int kSelBins = 4096;        // leading 12 bits of the float (sign is 0: 8 exponent + 3 mantissa bits ... see below)
constexpr int kSelCap = 4096;         // candidates sorted in LDS (32 KB of 64-bit composites)

// bin of a non-negative float: its 12 leading bits below the sign (exponent + 4 mantissa bits)
__device__ __forceinline__ unsigned sel_bin(unsigned bits) {
  const unsigned b = bits >> 19;
  return b < (unsigned)kSelBins ? b : (unsigned)kSelBins - 1;    // negative / NaN keys are outside the contract; no wild index
}

__global__ __launch_bounds__(kSelThreads) void sample_indices_kernel(const float *__restrict__ keys, long key_stride,
                                                                    const long *__restrict__ count, int n_sample,
                                                                    long *__restrict__ idx, int *__restrict__ overflow) {
  __shared__ unsigned hist[kSelBins];
  __shared__ unsigned long long cand[kSelCap];
  __shared__ unsigned s_thr, s_ncand;
  const int p = blockIdx.x, tid = threadIdx.x;
  const long n = count[p];
  const float *k = keys + (size_t)p * key_stride;
  long *out = idx + (size_t)p * n_sample;
  if (n <= n_sample) {                                   // with replacement (also n == 0: every index 0)
    for (int i = tid; i < n_sample; i += kSelThreads) out[i] = (long)floor((double)k[i] * (double)n);
    if (tid == 0) overflow[p] = 0;
    return;
  }
  for (int i = tid; i < kSelBins; i += kSelThreads) hist[i] = 0;
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  for (long i = tid; i < n; i += kSelThreads) atomicAdd(&hist[sel_bin(__float_as_uint(k[i]))], 1u);
  __syncthreads();
  if (tid == 0) {                                        // the bin holding the n_sample-th smallest key
    unsigned cum = 0, t = 0;
    for (; t < kSelBins; ++t) {
      cum += hist[t];
      if (cum >= (unsigned)n_sample) break;
    }
    s_thr = t;
    if (cum > (unsigned)kSelCap) s_ncand = 0xffffffffu;  // too many keys up to the threshold bin for the LDS sort
  }
  __syncthreads();
  const unsigned thr = s_thr;
  if (s_ncand == 0xffffffffu) {
    if (tid == 0) overflow[p] = 1;
    return;
  }
  for (int i = tid; i < kSelCap; i += kSelThreads) cand[i] = ~0ull;
  __syncthreads();
  for (long i = tid; i < n; i += kSelThreads) {
    const unsigned b = __float_as_uint(k[i]);
    if (sel_bin(b) <= thr) cand[atomicAdd(&s_ncand, 1u)] = ((unsigned long long)b << 32) | (unsigned long long)i;
  }
  __syncthreads();
  // bitonic sort of kSelCap 64-bit composites, ascending (the padding ~0 sorts last)
  for (int size = 2; size <= kSelCap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < kSelCap / 2; t += kSelThreads) {
        const int lo = 2 * t - (t & (stride - 1));        // index of the lower element of pair t at this stride
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = cand[lo], c = cand[hi];
        if ((a > c) == up) {
          cand[lo] = c;
          cand[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n_sample; i += kSelThreads) out[i] = (long)(cand[i] & 0xffffffffull);
  if (tid == 0) overflow[p] = 0;
}

// Masked colour crop of every surviving detection, resized to S x S and normalised (run_inference_custom.py:229-236:
// uint8 crop * uint8 mask, cv2.resize(INTER_LINEAR), ToTensor + Normalize).  The resize is OpenCV's fixed-point algorithm for
// CV_8U restated exactly (OpenCV 4.x modules/imgproc/src/resize.cpp; oracle/pem_pre.py cv2_resize_linear_u8 is the same
// statement in numpy): per axis  scale = 1. / (double(S) / n),  f = (float)((o + 0.5) * scale - 0.5),  s = floor(f),  f -= s;
// x axis: s < 0 -> (0, f = 0), s >= n - 1 -> (n - 1, f = 0); y axis: the two row indices are clipped instead; coefficients
// saturate_cast<short>((1 - f) * 2048), saturate_cast<short>(f * 2048) (round half to even, each on its own);
// t = S[sx] a0 + S[sx + 1] a1 per row (int32), dst = (((b0 (t0 >> 4)) >> 16) + ((b1 (t1 >> 4)) >> 16) + 2) >> 2.
// An exact 2:1 ratio on both axes takes OpenCV's area substitute (four-pixel sum + 2) >> 2, a 1:1 ratio is a copy.
// image (H,W,3) u8 RGB, m (P,H,W) u8, kept (M) i64, box (P,4) i64 -> out (M,3,S,S) f32, channel c = image channel 2 - c.
__device__ __forceinline__ void cv_linear_tap(int o, int S, long n, bool clamp_index, long &s, int &c0, int &c1) {
  const double inv = (double)S / (double)n;
  const double scale = 1.0 / inv;
  float f = (float)(((double)o + 0.5) * scale - 0.5);
  const float fl = floorf(f);
  s = (long)fl;
  f -= fl;
  if (clamp_index) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
  }
  c0 = (int)fminf(fmaxf(rintf((1.f - f) * 2048.f), -32768.f), 32767.f);
  c1 = (int)fminf(fmaxf(rintf(f * 2048.f), -32768.f), 32767.f);
}

__global__ void pem_crops_kernel(const unsigned char *__restrict__ image, const unsigned char *__restrict__ m,
                                 const long *__restrict__ kept, const long *__restrict__ box, int M, int H, int W, int S,
                                 int use_mask, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                                 float *__restrict__ out) {
  const size_t total = (size_t)M * 3 * S * S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S), c = (int)((i / ((size_t)S * S)) % 3), k = (int)(i / ((size_t)3 * S * S));
    const long p = kept[k];
    const long y1 = box[p * 4 + 0], y2 = box[p * 4 + 1], x1 = box[p * 4 + 2], x2 = box[p * 4 + 3];
    const long h = y2 - y1, w = x2 - x1;
    const int ch = 2 - c;
    const unsigned char *mp = m + (size_t)p * H * W;
    auto px = [&](long yy, long xx) -> int {                           // uint8 crop * uint8 mask of the reference
      const long y = y1 + yy, x = x1 + xx;
      const int v = (int)image[(y * W + x) * 3 + ch];
      return use_mask ? (mp[y * W + x] ? v : 0) : v;
    };
    int g;
    if (h == S && w == S) {
      g = px(oy, ox);
    } else if (h == 2 * (long)S && w == 2 * (long)S) {
      g = (px(2 * oy, 2 * ox) + px(2 * oy, 2 * ox + 1) + px(2 * oy + 1, 2 * ox) + px(2 * oy + 1, 2 * ox + 1) + 2) >> 2;
    } else {
      long sx, sy;
      int a0, a1, b0, b1;
      cv_linear_tap(ox, S, w, true, sx, a0, a1);
      cv_linear_tap(oy, S, h, false, sy, b0, b1);
      const long ya = min(max(sy, 0L), h - 1), yb = min(max(sy + 1, 0L), h - 1), xb = min(sx + 1, w - 1);
      const int t0 = px(ya, sx) * a0 + px(ya, xb) * a1;
      const int t1 = px(yb, sx) * a0 + px(yb, xb) * a1;
      g = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
      g = min(max(g, 0), 255);
    }
    const float mean = c == 0 ? mean0 : (c == 1 ? mean1 : mean2), sd = c == 0 ? std0 : (c == 1 ? std1 : std2);
    out[i] = ((float)g / 255.f - mean) / sd;
  }
}

}  // namespace s6d

using namespace s6d;

extern "C" int s6d_pem_sample_indices_f32(const float *keys, long key_stride, const int64_t *count, int P, int n_sample,
                                          int64_t *idx, int32_t *overflow, void *stream) {
  if (P < 0 || n_sample <= 0 || n_sample > kSelCap / 2 || key_stride < n_sample) return S6D_EINVAL;
  if (P == 0) return S6D_OK;
  if (!keys || !count || !idx || !overflow) return S6D_EINVAL;
  hipLaunchKernelGGL(sample_indices_kernel, dim3((unsigned)P), dim3(kSelThreads), 0, as_stream(stream), keys, key_stride,
                     (const long *)count, n_sample, (long *)idx, overflow);
  return launch_status();
}

extern "C" int s6d_pem_compact_cloud_f32(const unsigned char *m, const float *depth, const int64_t *box, const unsigned char *ok,
                                         int P, int H, int W, float fx, float fy, float cx, float cy, long cap, int32_t *choose,
                                         float *cloud, int64_t *n, void *stream) {
  if (P < 0 || H <= 0 || W <= 0 || cap <= 0 || fx == 0.f || fy == 0.f) return S6D_EINVAL;
  if (P == 0) return S6D_OK;
  if (!m || !depth || !box || !ok || !choose || !cloud || !n) return S6D_EINVAL;
  hipLaunchKernelGGL(compact_cloud_kernel, dim3((unsigned)P), dim3(kCmpThreads), 0, as_stream(stream), m, depth,
                     (const long *)box, ok, H, W, fx, fy, cx, cy, cap, choose, cloud, (long *)n);
  return launch_status();
}

extern "C" int s6d_pem_radius_filter_f32(const float *center, const double *limit, int P, long cap, int32_t *choose,
                                         float *cloud, int64_t *n, void *stream) {
  if (P < 0 || cap <= 0) return S6D_EINVAL;
  if (P == 0) return S6D_OK;
  if (!center || !limit || !choose || !cloud || !n) return S6D_EINVAL;
  hipLaunchKernelGGL(radius_filter_kernel, dim3((unsigned)P), dim3(kCmpThreads), 0, as_stream(stream), center, limit, cap,
                     choose, cloud, (long *)n);
  return launch_status();
}

extern "C" int s6d_pem_mask_boxes_u8(const unsigned char *mask, const float *depth, int P, int H, int W, long min_points,
                                     unsigned char *m, int64_t *cnt, unsigned char *ok, int64_t *box, void *stream) {
  if (P < 0 || H <= 0 || W <= 0 || (long)H * W > 0x7fffffffL) return S6D_EINVAL;
  if (P == 0) return S6D_OK;
  if (!mask || !depth || !m || !cnt || !ok || !box) return S6D_EINVAL;
  hipLaunchKernelGGL(mask_boxes_kernel, dim3((unsigned)P), dim3(kCmpThreads), 0, as_stream(stream), mask, depth, H, W, min_points,
                     m, (long *)cnt, ok, (long *)box);
  return launch_status();
}

extern "C" int s6d_pem_crops_f32(const unsigned char *image, const unsigned char *m, const int64_t *kept, const int64_t *box,
                                 int M, int H, int W, int S, int use_mask, const float *mean3_host, const float *std3_host,
                                 float *out, void *stream) {
  if (M < 0 || H <= 0 || W <= 0 || S <= 0) return S6D_EINVAL;
  if (M == 0) return S6D_OK;
  if (!image || !m || !kept || !box || !mean3_host || !std3_host || !out) return S6D_EINVAL;
  const size_t total = (size_t)M * 3 * S * S;
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(pem_crops_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), image, m, (const long *)kept,
                     (const long *)box, M, H, W, S, use_mask, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0],
                     std3_host[1], std3_host[2], out);
  return launch_status();
}
