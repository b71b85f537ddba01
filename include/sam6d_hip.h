/*
 * sam6d_hip.h -- C ABI of libsam6d_hip.so: the MI355X (gfx950) hot path of SAM-6D.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the name ends in _host.  Tensors are dense, row-major, in the layout the
 *     comment gives.  No ownership is taken; outputs are caller-allocated.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     entry points only enqueue work on that stream; none synchronises.
 *   - return value: S6D_OK (0) or a negative S6D_E* code; s6d_strerror() names
 *     it.  Unlike the reference extension (cuda_utils.h:35-44: launch failure
 *     -> exit(-1)) a failed launch is reported to the caller.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to SAM-6D/ in the reference tree).
 */
#ifndef SAM6D_HIP_H
#define SAM6D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S6D_OK 0
#define S6D_EINVAL (-1)   /* bad size / null pointer / unsupported shape */
#define S6D_ELAUNCH (-2)  /* hipLaunchKernel / hipGetLastError failure    */
#define S6D_EUNSUPPORTED (-3)

/* Bumped whenever a signature in this header changes; the loader (sam6d_amd/_lib.py) refuses a library whose
 * s6d_version() differs from the header it was written against (a stale .so fails at load, not at a call). */
#define S6D_ABI_VERSION 124
int s6d_version(void);
/* Upper bound on the workgroups of the persistent kernels (the 14 x 14 window attention walks its (window, head) items with one
 * workgroup per CU); 0 = one per CU of the device.  Process-wide.  Replaces the environment lookups the launch path made until
 * round 4; used by the tests to make few workgroups walk many items. */
int s6d_set_persistent_grid_limit(int max_workgroups);
/* Which form of the bf16 / f16 GEMM kernel serves the shapes both forms cover (csrc/s6d_gemm.hip: eight waves, 128 x 64 wave tiles;
 * csrc/s6d_gemm4.hip: four waves, 128 x 128 wave tiles, accumulators in the accumulator register file): 0 = the library's choice per
 * shape (the default: the eight-wave form, measured 0.3 % ahead inside the benched step), 64 = always the eight-wave form, 128 = the
 * four-wave form wherever it applies (bf16 / f16, N % 256 == 0, M % 256 == 0, K >= 128).  Both forms give the same bits (the same products in the same order per
 * accumulator); the switch exists for A/B measurements and the parity tests.  Process-wide; returns S6D_EINVAL for other values. */
int s6d_set_gemm_wave_tile(int columns);
/* 1 (default): a plain or GELU bf16 / f16 GEMM launch that would put fewer than 160 tiles of 256 x 256 on the chip takes the 256 x 128
 * tile kernel (two independent workgroups per CU) -- the PEM ViT-B's 6304 x 768 products are 75 tiles otherwise.  0: always the
 * 256 x 256 kernel.  2: as 1, and the residual + row-statistics epilogue (the ViT-H's proj / lin2 at one frame) as well -- built and
 * bit-equal, measured 6 % slower for the one-frame encoder, hence not part of the default.  The same products in the same order per
 * element in every mode: the same bits.  Process-wide; other values: S6D_EINVAL. */
int s6d_set_gemm_small_tile(int enable);
const char *s6d_strerror(int code);
/* last HIP error string seen by this thread (empty if none) */
const char *s6d_last_hip_error(void);

/* ---------------------------------------------------------------- PointNet++ ops
 * Replace pybind module pointnet2._ext
 * (Pose_Estimation_Model/model/pointnet2/_ext_src/src/bindings.cpp:11-24). */

/* furthest_point_sampling(points (B,N,3) f32, nsamples) -> idx (B,M) i32
 * ref: _ext_src/src/sampling.cpp:70-91, sampling_gpu.cu:74-178.
 * `tmp` (B*N f32 scratch) is only needed when N > 4096 (may be NULL otherwise).
 * Bit-exact with the reference's selection rule incl. its tie-break (ties go to the
 * smallest bit-reversed slot k mod opt_n_threads(N) of the shared-memory tree, then
 * to the lowest k). */
int s6d_fps_f32(const float *xyz, int B, int N, int M, float *tmp, int32_t *idx, void *stream);

/* gather_points(points (B,C,N) f32, idx (B,M) i32) -> out (B,C,M)
 * ref: _ext_src/src/sampling.cpp:18-43, sampling_gpu.cu:13-25. */
int s6d_gather_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int M,
                          float *out, void *stream);

/* ball_query(new_xyz (B,M,3), xyz (B,N,3), radius, nsample) -> idx (B,M,nsample) i32
 * ref: _ext_src/src/ball_query.cpp:13-37, ball_query_gpu.cu:14-49 (first-hit fill,
 * strict d2 < r*r, zeros when a centre has no neighbour). */
int s6d_ball_query_f32(const float *new_xyz, const float *xyz, int B, int N, int M, float radius,
                       int nsample, int32_t *idx, void *stream);

/* group_points(points (B,C,N) f32, idx (B,M,S) i32) -> out (B,C,M,S)
 * ref: _ext_src/src/group_points.cpp:14-38, group_points_gpu.cu:13-33. */
int s6d_group_points_f32(const float *points, const int32_t *idx, int B, int C, int N, int M, int S,
                         float *out, void *stream);

/* Row gather used by the product modules instead of transpose+gather_points+transpose
 * (Pose_Estimation_Model/utils/model_utils.py:53-66, model/transformer.py:651-658):
 * src (B,N,C) f32, idx (B,M) i32 -> out (B,M,C). */
int s6d_gather_rows_f32(const float *src, const int32_t *idx, int B, int N, int C, int M, float *out,
                        void *stream);

/* Sum of each segment's rows in ROW ORDER, float32 accumulator: numpy's add.reduce over axis 0 of a C-ordered
 * (n, C) array, i.e. the numerator of np.mean(cloud, axis=0) in the PEM pre-processing
 * (Pose_Estimation_Model/run_inference_custom.py:214, provider/bop_test_dataset.py:131).
 * x (N,C) f32, start/count (P) i64 (rows [start, start+count) of x), 2 <= C <= 4 -> out (P,C) f32 (0 for empty).
 * (C = 1 is refused: with a single column the reduced axis is the contiguous one and numpy sums it pairwise.) */
int s6d_segment_seq_sum_f32(const float *x, const int64_t *start, const int64_t *count, int P, int C, float *out,
                            void *stream);

/* Point sampler of the PEM pre-processing in its defined form (sam6d_amd/pem/preprocess.py header; reference draws:
 * Pose_Estimation_Model/run_inference_custom.py:224-229, provider/bop_test_dataset.py:140-145).
 * keys (P, key_stride) f32 non-negative uniforms, count (P) i64 candidate points per detection (count <= key_stride),
 * 1 <= n_sample <= 2048 <= key_stride -> idx (P, n_sample) i64: count <= n_sample: floor(key_i * count) (with
 * replacement); else positions of the n_sample smallest (key, position) pairs in ascending order.  overflow (P) i32 is
 * set to 1 for a detection whose keys are too duplicated for the in-LDS selection (its idx row is then not written). */
int s6d_pem_sample_indices_f32(const float *keys, long key_stride, const int64_t *count, int P, int n_sample,
                               int64_t *idx, int32_t *overflow, void *stream);

/* m = mask AND depth > 0, its pixel count, ok = count > min_points, and the square crop box of get_bbox
 * (Pose_Estimation_Model/utils/data_utils.py:126-160; run_inference_custom.py:204-208) per detection, in one pass.
 * mask (P,H,W) u8 (non-zero = set), depth (H,W) f32 -> m (P,H,W) u8, cnt (P) i64, ok (P) u8, box (P,4) i64 [y1,y2,x1,x2]
 * (the whole-frame box for a detection that is not ok). */
int s6d_pem_mask_boxes_u8(const unsigned char *mask, const float *depth, int P, int H, int W, long min_points,
                          unsigned char *m, int64_t *cnt, unsigned char *ok, int64_t *box, void *stream);

/* Masked crop pixels of every detection, in row-major crop order, with their back-projected points
 * (run_inference_custom.py:209-213, utils/data_utils.py:92-110).  m (P,H,W) u8 = mask AND depth > 0, depth (H,W) f32,
 * box (P,4) i64 [y1,y2,x1,x2], ok (P) u8 (0: detection skipped, n = 0) -> choose (P,cap) i32 crop-flat indices,
 * cloud (P,cap,3) f32, n (P) i64; cap >= the largest crop area (min(H,W)^2 always suffices). */
int s6d_pem_compact_cloud_f32(const unsigned char *m, const float *depth, const int64_t *box, const unsigned char *ok, int P,
                              int H, int W, float fx, float fy, float cx, float cy, long cap, int32_t *choose, float *cloud,
                              int64_t *n, void *stream);
/* In place: keep, in order, the points with (double)|cloud - center[p]| < limit[p] (run_inference_custom.py:214-221).
 * center (P,3) f32, limit (P) f64, choose / cloud / n as above (n is updated). */
int s6d_pem_radius_filter_f32(const float *center, const double *limit, int P, long cap, int32_t *choose, float *cloud,
                              int64_t *n, void *stream);

/* Masked colour crops of the surviving detections, resized to S x S and normalised (run_inference_custom.py:231-236 in
 * the defined form of sam6d_amd/pem/preprocess.py: float32 bilinear, rounded to a grey level, ToTensor + Normalize).
 * image (H,W,3) u8 RGB, m (P,H,W) u8, kept (M) i64 detection indices, box (P,4) i64, mean / std: 3 floats each on the HOST
 * -> out (M,3,S,S) f32 with channel c = image channel 2 - c (the reference's [:, :, ::-1]). */
int s6d_pem_crops_f32(const unsigned char *image, const unsigned char *m, const int64_t *kept, const int64_t *box, int M,
                      int H, int W, int S, int use_mask, const float *mean3_host, const float *std3_host, float *out,
                      void *stream);

/* ---------------------------------------------------------------- PEM pose solvers
 * Replace the library-op chains of Pose_Estimation_Model/utils/model_utils.py. */

/* R = V diag(1,1,det(V U^T)) U^T for H = U S V^T, n matrices: H (n,3,3) f32 -> R (n,3,3) f32.
 * ref: weighted_procrustes, utils/model_utils.py:341-347 (torch.svd + det fix). */
int s6d_rot_from_h_f32(const float *H, int n, float *R, void *stream);

/* weighted_procrustes for one point set per instance (utils/model_utils.py:287-363 as compute_fine_Rt calls it, :268-271):
 * src, ref (B,N,3) f32, weights (B,N) f32 (values below weight_thresh count as 0; normalised by sum + eps) -> R (B,3,3),
 * t (B,3) with ref ~ src R^T + t.  One workgroup per instance, fixed summation order in double: an instance's result
 * does not depend on B (the library op chain it replaces picked reduction / bmm configurations by batch size). */
int s6d_weighted_procrustes_f32(const float *src, const float *ref, const float *weights, int B, int N, float weight_thresh,
                                float eps, float *R, float *t, void *stream);

/* Pose hypotheses of compute_coarse_Rt (utils/model_utils.py:216-231).
 * pts1 (B,N1,3), pts2 (B,N2,3) f32; pair (B,n_hyp,3) i32 = searchsorted() bins, decoded as
 * (i1, i2) = (pair / N2, pair % N2), clamped like the reference.  Outputs R (B,n_hyp,3,3),
 * t (B,n_hyp,3), dis (B,n_hyp) = mean_i |(p1_i - t) R - p2_i|. */
int s6d_pose_hypotheses_f32(const float *pts1, const float *pts2, const int32_t *pair, int B, int N1,
                            int N2, int n_hyp, float *R, float *t, float *dis, void *stream);

/* Coarse hypothesis sampling: dual softmax of atten (B,M1,M2), background labels, (score[1:,1:])^1.5, float64-accumulated prefix
 * sums normalised by (total + 1e-8), lower-bound search of the caller's uniforms rand_u (B,n_u) -> pair (B,n_u) i32 (flat bin
 * index i1 * (M2-1) + i2, or (M1-1)(M2-1) past the end, as torch.searchsorted) and w1 (B,M1-1) f32 (label1 > 0).
 * One workgroup per instance, the 196 x 196 distribution stays in LDS ((M1-1)(M2-1) * 4 B <= ~155 KB, M2 <= 256).
 * ref: compute_coarse_Rt, utils/model_utils.py:203-219. */
int s6d_coarse_sample_f32(const float *atten, const float *rand_u, int B, int M1, int M2, int n_u, int32_t *pair, float *w1,
                          void *stream);

/* The k smallest of n residuals per instance in ascending (value, index) order and the rows of Rs (B,n,3,3), ts (B,n,3) they
 * select -> Rk (B,k,3,3), tk (B,k,3), idx (B,k) i32.  ref: torch.topk(dis, 300, largest=False) + gathers, model_utils.py:233-235. */
int s6d_smallest_k_f32(const float *dis, const float *Rs, const float *ts, int B, int n, int k, float *Rk, float *tk,
                       int32_t *idx, void *stream);

/* score_p = sum(w1) / (sum_n dmin[b,p,n] w1[b,n] + 1e-8); R, t of the first arg-max.  dmin (B,P,N), w1 (B,N), Rk (B,P,3,3),
 * tk (B,P,3) -> R (B,3,3), t (B,3).  ref: model_utils.py:240-246. */
int s6d_hypothesis_select_f32(const float *dmin, const float *w1, const float *Rk, const float *tk, int B, int P, int N,
                              float *R, float *t, void *stream);

/* dmin[b,p,n] = min_m |(pts[b,n] - t[b,p]) R[b,p] - model[b,m]|
 * pts (B,N,3), R (B,P,3,3), t (B,P,3), model (B,Nm,3) -> dmin (B,P,N).
 * ref: compute_coarse_Rt :234-239 (P = 300) and compute_fine_Rt :272-275 (P = 1). */
int s6d_min_dist_f32(const float *pts, const float *R, const float *t, const float *model, int B, int N,
                     int P, int Nm, float *dmin, void *stream);

/* ---------------------------------------------------------------- PEM point transformer */

/* RPE attention core: softmax_m((q.k + q~.e + qb) * scale) v, heads = 4, C = 256.
 * q,k,v (B,N,C) f32 (projected, heads interleaved "(h c)"); qt (B,4,N,C) = W_p[h]^T q[b,h,n];
 * qb (B,4,N) = q[b,h,n].b_p[h]; embed (B,N,N,C) geometric structure embedding -> out (B,N,C).
 * ref: RPEMultiHeadAttention.forward, Pose_Estimation_Model/model/transformer.py:368-406
 * (which materialises proj_p(embed) (B,4,N,N,64); here embed is streamed once). */
int s6d_rpe_attention_f32(const float *q, const float *k, const float *v, const float *qt, const float *qb,
                          const float *embed, int B, int N, int C, int heads, float scale, float *out,
                          void *stream);

/* The same core with q | k | v | q~ | qb taken as column blocks of ONE projection output proj (B,N,ld) f32 (offsets in floats, multiples
 * of 4; q~ is 4 x 256 wide: head h at qt_off + 256 h; qb 4 wide): the caller folds W_p into the projection's weights
 * (q~_h = x (W_q,h^T W_p,h) + b_q,h W_p,h; qb_h = x (W_q,h^T b_p,h) + b_q,h . b_p,h), so `W_p^T q` is not a pass of its own.
 * ref: as s6d_rpe_attention_f32 (transformer.py:368-406). */
int s6d_rpe_attention_packed_f32(const float *proj, long ld, int q_off, int k_off, int v_off, int qt_off, int qb_off,
                                 const float *embed, int B, int N, int C, int heads, float scale, float *out, void *stream);
/* ... with the embedding (B,N,N,C) stored in IEEE half (s6d_geo_embedding_f16): widened in registers, float32 products and sums. */
int s6d_rpe_attention_packed_e16_f32(const float *proj, long ld, int q_off, int k_off, int v_off, int qt_off, int qb_off,
                                     const void *embed_f16, int B, int N, int C, int heads, float scale, float *out, void *stream);
int s6d_rpe_attention_strided_e16_f32(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv, const float *qt,
                                      const float *qb, const void *embed_f16, int B, int N, int C, int heads, float scale, float *out,
                                      void *stream);
/* The same with row strides ldq / ldk / ldv (floats, multiples of 4) for q / k / v: the column blocks of one q | k | v projection
 * output are attended without copies. */
int s6d_rpe_attention_strided_f32(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv, const float *qt,
                                  const float *qb, const float *embed, int B, int N, int C, int heads, float scale, float *out,
                                  void *stream);

/* ---------------------------------------------------------------- SAM image encoder */

/* Fused (windowed or global) multi-head attention with decomposed relative-position bias.
 * qkv (B,H,W,3,nh,hd) bf16 = output of the qkv Linear on the REAL tokens (no window padding);
 * qkv_bias (3,nh,hd) bf16 = q/k/v of out-of-image window slots (the reference pads zeros after
 * norm1, so those tokens equal the qkv bias and take part as keys); rel_h, rel_w (2S-1,hd) bf16
 * (S = window, or H for window == 0 -> global attention; may both be NULL = no bias);
 * out (B,H,W,nh*hd) bf16.  softmax(scale*q.k + q.rel_h[qy-ky+S-1] + q.rel_w[qx-kx+S-1]) v.
 * hd in {64, 80}.
 * ref: segment_anything/modeling/image_encoder.py Block.forward :166-182, Attention.forward
 * :224-240, add_decomposed_rel_pos :325-361, window_partition/unpartition :243-289. */
int s6d_win_attention_bf16(const void *qkv, const void *qkv_bias, const void *rel_h, const void *rel_w, int B,
                           int H, int W, int num_heads, int head_dim, int window, float scale, void *rel_scratch,
                           void *out, void *stream);
/* The same attention on a head-major q/k/v tensor (head_major = 1: qkv is (3, num_heads, B*H*W, head_dim), the column-block output of
 * s6d_gemm_bf16_cblk; head_major = 0: the raw Linear output (B,H,W,3,num_heads,head_dim) as above). */
int s6d_win_attention_layout_bf16(const void *qkv, int head_major, const void *qkv_bias, const void *rel_h, const void *rel_w,
                                  int B, int H, int W, int num_heads, int head_dim, int window, float scale,
                                  void *rel_scratch, void *out, void *stream);
/* The same with the padded table copies made ahead of time: s6d_win_attention_pad_rel_bf16 writes them (s6d_win_attention_scratch_bytes
 * bytes) from rel_h / rel_w once per weight version, s6d_win_attention_prepadded_bf16 is s6d_win_attention_layout_bf16 reading them --
 * no padding launch in front of every attention launch. */
int s6d_win_attention_pad_rel_bf16(const void *rel_h, const void *rel_w, int H, int window, int head_dim, void *rel_padded, void *stream);
int s6d_win_attention_prepadded_bf16(const void *qkv, int head_major, const void *qkv_bias, const void *rel_padded, int B, int H, int W,
                                     int num_heads, int head_dim, int window, float scale, void *out, void *stream);
/* bytes of `rel_scratch` (zero-padded copies of the two tables; may be NULL when rel_h == NULL) */
long s6d_win_attention_scratch_bytes(int H, int window, int head_dim);

/* x_out = x + delta (skipped when delta == NULL: x_out may be NULL), y_out = LayerNorm_C(x_out)*gamma+beta.
 * x, delta, x_out, y_out (rows,C) bf16; gamma, beta (C) f32; fp32 statistics; C % 8 == 0, C <= 2048.
 * ref: the residual adds and norm1/norm2 of Block.forward, segment_anything/modeling/image_encoder.py:166-182. */
int s6d_add_layernorm_bf16(const void *x, const void *delta, const float *gamma, const float *beta, float eps,
                           long rows, int C, void *x_out, void *y_out, void *stream);
/* LayerNorm of bf16 rows with an fp32 result (no residual): the last LayerNorm2d of the SAM neck, whose output is the image
 * embedding handed to the mask decoder in float32 (segment_anything/modeling/common.py:31-43 applied channels-last). */
int s6d_layernorm_bf16_f32(const void *x, const float *gamma, const float *beta, float eps, long rows, int C, float *y_f32,
                           void *stream);

/* C = epilogue(A W^T + bias): the nn.Linear layers of the ViTs with the bias and the activation folded into the GEMM.
 * A (M,K) bf16, row stride lda; W (N,K) bf16 = nn.Linear.weight, row stride ldw; bias (N) f32 or NULL; C (M,N) bf16, row
 * stride ldc (strides in elements, multiples of 8; 16-byte aligned bases).  epilogue: 0 = none, 1 = exact (erf) GELU.
 * N % 128 == 0 (N % 256 == 0 takes the 256 x 256-tile kernel, otherwise the 256 x 128-tile one), K % 64 == 0, any M.  max_blocks: workgroups to launch (<= 0: one persistent workgroup per CU, 256).
 * fp32 accumulation on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), one rounding to bf16 at the end.
 * ref: segment_anything/modeling/common.py:13-28 (MLPBlock: lin1 -> GELU -> lin2), image_encoder.py:224-240 (qkv, proj),
 * :90-104 (neck 1x1 conv); the same statements in timm's ViT-B (Pose_Estimation_Model/model/feature_extraction.py:17-35)
 * and DINOv2 ViT-L (Instance_Segmentation_Model/model/layers/{attention,mlp}.py). */
int s6d_gemm_bf16(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M, int N,
                  int K, int epilogue, int max_blocks, void *stream);
/* The same product with the output in COLUMN BLOCKS: block j (columns [j col_block, (j+1) col_block)) is a contiguous (M, col_block)
 * matrix at C + j * M * col_block (ldc unused).  col_block % 8 == 0, N % col_block == 0, N % 256 == 0.  The qkv projection of a ViT
 * block writes q / k / v head-major this way (col_block = head_dim) for s6d_win_attention_layout_bf16(head_major = 1). */
int s6d_gemm_bf16_cblk(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M, int N, int K,
                       int epilogue, int col_block, int max_blocks, void *stream);
/* IEEE-half (float16) builds of three kernels, for the PEM's ViT-B (Pose_Estimation_Model/model/feature_extraction.py:17-35 on
 * timm's VisionTransformer): same signatures and layouts as their bf16 namesakes, elements are IEEE binary16.  Half's 11-bit
 * significand keeps the extractor's features within 1e-3 of the fp32 extractor's (bf16: 7.6e-3), which the matcher's 1e-3 mm
 * translation bar needs; the matrix rate is the bf16 one.  s6d_gemm_f16: N % 256 == 0, epilogue 0 / 1. */
int s6d_gemm_f16(const void *A, long lda, const void *W, long ldw, const float *bias, void *C, long ldc, int M, int N, int K,
                 int epilogue, int max_blocks, void *stream);
int s6d_add_layernorm_f16(const void *x, const void *delta, const float *gamma, const float *beta, float eps, long rows, int C,
                          void *x_out, void *y_out, void *stream);
int s6d_seq_attention_f16(const void *qkv, int B, int N, int num_heads, int head_dim, float scale, void *out, void *stream);
/* The same on a strided q / k / v tensor: element (sequence b, token n, which, head h, d) at qkv + (b N + n) tok_stride +
 * which which_stride + h head_stride + d (elements; strides % 8 == 0).  Token-major = s6d_seq_attention_*; head-major
 * ((3, nh, B N, hd), the column-block output of s6d_gemm_bf16_cblk / _lnfold: tok_stride = hd, head_stride = B N hd, which_stride =
 * nh B N hd) makes the K / V rows of a (sequence, head) one contiguous run.  out stays (B, N, nh, hd).
 * ref: as s6d_seq_attention_bf16 (timm Attention.forward; Instance_Segmentation_Model/model/layers/attention.py:29-62). */
int s6d_seq_attention_strided_bf16(const void *qkv, long tok_stride, long which_stride, long head_stride, int B, int N,
                                   int num_heads, int head_dim, float scale, void *out, void *stream);
int s6d_seq_attention_strided_f16(const void *qkv, long tok_stride, long which_stride, long head_stride, int B, int N,
                                  int num_heads, int head_dim, float scale, void *out, void *stream);

/* The residual add and the LayerNorm of a transformer block folded into the GEMMs on either side of them
 * (segment_anything/modeling/image_encoder.py:166-182: `x = shortcut + x`, `x = x + self.mlp(self.norm2(x))`, `self.norm1(x)`;
 * timm / DINOv2 blocks alike).  All: bf16, N % 256 == 0.
 *
 * s6d_gemm_bf16_res: C = bf16(A W^T + bias + R), R (M,N) bf16 with row stride ldr (R may be C: in place).  The accumulators start at
 *   bias + residual (a tile's residual is read one tile ahead, during the previous tile's epilogue), so the sum is formed in fp32
 *   and rounded once.  stats_partial (optional, else NULL): [N / 32][2][M] floats, for every row and 32-column group the sum and
 *   the sum of squared deviations from the group mean of the fp32 results -- the input of s6d_ln_stats_finalize.
 * s6d_ln_stats_finalize: row_stats[m] = (mean, sigma = sqrt(var + eps)) over groups * group_size columns from such partials
 *   (pairwise-exact combination, fixed order).  s6d_row_stats_bf16: the same statistics computed from a bf16 matrix (two-pass).
 * s6d_gemm_bf16_lnfold: C = act(LN(A) W^T + b) without materialising LN(A):  W is the FOLDED weight gamma o W (bf16), col_sums[n] =
 *   sum_k W_folded[n,k], bias[n] = b[n] + sum_k beta[k] W[n,k], row_stats from the calls above;
 *   C[m,n] = act(((A W_folded^T)[m,n] + sigma_m bias[n] - mean_m col_sums[n]) / sigma_m).  gelu 0 / 1; col_block as
 *   s6d_gemm_bf16_cblk. */
int s6d_gemm_bf16_res(const void *A, long lda, const void *W, long ldw, const float *bias, const void *R, long ldr,
                      float *stats_partial, void *C, long ldc, int M, int N, int K, int max_blocks, void *stream);
int s6d_ln_stats_finalize(const float *stats_partial, int groups, int group_size, long M, float eps, float *row_stats, void *stream);
int s6d_row_stats_bf16(const void *x, long ldx, long M, int C, float eps, float *row_stats, void *stream);
int s6d_gemm_bf16_lnfold(const void *A, long lda, const float *row_stats, const void *W, long ldw, const float *col_sums,
                         const float *bias, void *C, long ldc, int M, int N, int K, int gelu, int col_block, int max_blocks,
                         void *stream);

/* ---------------------------------------------------------------- fp8 ViT path (BASELINE configs[4]; never the headline)
 * C = act(A W^T + bias) with OCP fp8 (e4m3fn) operands: A (M,K) and W (N,K) bytes, row strides lda / ldw (multiples of 16),
 * a_scale (M) / w_scale (N): one E8M0 byte per row -- the real value of an element is q * 2^(byte - 127) -- i.e. per-token
 * activation scales and per-output-channel weight scales, restricted to powers of two so that they ride in the block-scale
 * operands of v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 instruction's product per cycle).  fp32 accumulation, bias (N)
 * f32 or NULL, epilogue 0 / 1 (exact GELU), C (M,N) bf16.  N % 256 == 0, K % 128 == 0.
 * There is no fp8 path in the reference (its nearest precision hook: Instance_Segmentation_Model/configs/machine/trainer/
 * local.yaml:9, Lightning precision 16); the Linear statements are those of s6d_gemm_bf16. */
int s6d_gemm_fp8(const void *A, long lda, const unsigned char *a_scale, const void *W, long ldw, const unsigned char *w_scale,
                 const float *bias, void *C, long ldc, int M, int N, int K, int epilogue, int max_blocks, void *stream);
/* MX forms of the fp8 GEMM (round 4; BASELINE configs[4], lin1 -> lin2 of a ViT block without a bf16 round trip).
 * s6d_gemm_fp8_gelu_mx: C8 = e4m3(GELU(A W^T + bias)) (M,N) bytes with row stride ldc BYTES (ldc % 16 == 0) + c_scale [M][N / 32]:
 *   one E8M0 byte per row and 32 columns, scale = 2^e with the smallest e for which the block's amax / 2^e <= 448 (the rule of
 *   s6d_layernorm_fp8, per block).  A / a_scale / W / w_scale / bias as s6d_gemm_fp8.
 * s6d_gemm_fp8_mxa: A (M,K) e4m3 bytes with a_mx [M][K / 32] E8M0 bytes (what s6d_gemm_fp8_gelu_mx wrote; 4-byte aligned), W with
 *   one scale per output channel -> act(A W^T + bias) (M,N) bf16.  N % 256 == 0, K % 128 == 0; a_mx must be readable for
 *   ceil(M / 256) * 256 rows (the scale dwords of a whole row tile are fetched; rows past M may hold anything).
 * The matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4) applies a lane's scale byte to its own 32-k block and op_sel picks
 * the byte: the MX scales ride in the hardware operand, the accumulators hold the true product. */
int s6d_gemm_fp8_gelu_mx(const void *A, long lda, const unsigned char *a_scale, const void *W, long ldw, const unsigned char *w_scale,
                         const float *bias, void *C8, long ldc, unsigned char *c_scale, int M, int N, int K, int max_blocks,
                         void *stream);
int s6d_gemm_fp8_mxa(const void *A, long lda, const unsigned char *a_mx, const void *W, long ldw, const unsigned char *w_scale,
                     const float *bias, void *C, long ldc, int M, int N, int K, int epilogue, int max_blocks, void *stream);
/* LayerNorm(x) (rows,C) bf16 -> fp8 e4m3 rows y8 (rows,C) + one E8M0 scale byte per row (the A operand of s6d_gemm_fp8):
 * fp32 statistics, scale 2^e with the smallest e for which amax / 2^e <= 448, round to nearest even.  C % 8 == 0, C <= 2048. */
int s6d_layernorm_fp8(const void *x, const float *gamma, const float *beta, float eps, long rows, int C, void *y8,
                      unsigned char *yscale, void *stream);
/* The same with the residual add folded in (x_out = bf16(x + delta), then LayerNorm of x_out), as s6d_add_layernorm_bf16. */
int s6d_add_layernorm_fp8(const void *x, const void *delta, const float *gamma, const float *beta, float eps, long rows, int C,
                          void *x_out, void *y8, unsigned char *yscale, void *stream);

/* ---------------------------------------------------------------- ISM proposal-vs-template scoring */

/* clamp(cosine(query_p, ref_r), 0, 1): query (P,C), ref (R = O*T, C) f32 -> out (P,R) f32; C % 16 == 0.
 * ref: PairwiseSimilarity.forward, Instance_Segmentation_Model/model/loss.py:27-44. */
int s6d_pairwise_cosine_f32(const float *query, const float *ref, int P, int R, int C, float *out, void *stream);

/* Per proposal: mean of the top-k template scores per object, arg-max object, its score and the best
 * template of that object.  scores (P,O,T) f32 -> best_score (P) f32, best_obj (P) i32, best_tmpl (P) i32.
 * ref: compute_semantic_score ('avg_5': topk = 5) + best_template_pose, model/detector.py:260-296,198-207. */
int s6d_semantic_select_f32(const float *scores, int P, int O, int T, int topk, float *best_score,
                            int32_t *best_obj, int32_t *best_tmpl, void *stream);

/* Appearance score and visible ratio from ONE similarity pass.  query (S,N1,C) f32; refstore (O,T,N2,C) f32
 * indexed by obj[s], tmpl[s] (i32) -- no gathered copy; workspace: s6d_patch_scores_workspace_floats() floats.
 * appe[s] = clamp(sum_i max_j sim_ij / (#rows with non-zero element sum + 1e-6), 0, 1)
 * ratio[s] = #cols(max_i sim_ij > thred) / (#cols(max_i sim_ij != 0) + 1e-6)
 * ref: MaskedPatch_MatrixSimilarity.compute_straight / compute_visible_ratio, model/loss.py:52-76. */
int s6d_patch_scores_f32(const float *query, const float *refstore, const int32_t *obj, const int32_t *tmpl,
                         int S, int N1, int N2, int C, int T, float thred, float *workspace, float *appe,
                         float *ratio, void *stream);
long s6d_patch_scores_workspace_floats(int S, int N1, int N2);
/* The same with the selected proposals named by index: query (P,N1,C) is the UN-gathered descriptor tensor of all proposals of the
 * launch group and row s of the launch reads query[qsel[s]] (qsel (S) i32; NULL = s6d_patch_scores_f32).  Replaces the
 * `query_appe_descriptors[idx_selected]` copy of model/detector.py:341-349 (1 MB per selected proposal at 256 x 1024 floats). */
int s6d_patch_scores_sel_f32(const float *query, const int32_t *qsel, const float *refstore, const int32_t *obj,
                             const int32_t *tmpl, int S, int N1, int N2, int C, int T, float thred, float *workspace, float *appe,
                             float *ratio, void *stream);

/* Mean back-projected 3-D point of each mask: masks (S,H,W) f32, depth (H,W) f32 -> out (S,3) f32 [m], with
 * Z = depth * depth_scale / 1000 (the reference's contract: depth in millimetres at depth_scale 1).
 * K: the 3x3 row-major float64 camera matrix IN DEVICE MEMORY (read by the kernel: no host copy per frame).
 * ref: Calculate_the_query_translation, model/detector.py:234-246 +
 * depth_image_to_pointcloud_translate_torch, utils/trimesh_utils.py:77-105 (float64 X/Y, float32 Z).
 * The three sums are taken in the order of ATen's CPU `sum` (cascade_sum with AVX2 vectors: 32 float / 16 double columns,
 * 16-row cascade levels), so that the result carries the bits of the reference's CPU run (oracle/aten_sum.py). */
int s6d_masked_depth_mean_f32(const float *masks, const float *depth, int S, int H, int W, float depth_scale,
                              const double *K, void *workspace, float *out, void *stream);
long s6d_masked_depth_mean_workspace_bytes(int S, int H, int W);   /* W % 4 == 0, H*W < 2^24 */
/* Several frames in one launch: depth (F,H,W), K (F,3,3) f64, frame (S) i32 = the frame of every mask (NULL: one frame). */
int s6d_masked_depth_mean_frames_f32(const float *masks, const float *depth, const int32_t *frame, int S, int H, int W,
                                     float depth_scale, const double *K, void *workspace, float *out, void *stream);
/* ... and with the selected masks named by index: masks (P,H,W) holds every proposal of the launch group, mask s of the launch is
 * masks[msel[s]] (msel (S) i32; NULL: masks[s]).  Replaces the `masks[idx_selected]` copy of model/detector.py:351-353
 * (1.2 MB per selected proposal at 480 x 640).  Workspace as for S masks. */
int s6d_masked_depth_mean_sel_f32(const float *masks, const int32_t *msel, const float *depth, const int32_t *frame, int S, int H,
                                  int W, float depth_scale, const double *K, void *workspace, float *out, void *stream);

/* Template projection: uv[s,i] = clamp(trunc(K (R_tmpl[s] p_i + t_s))), bbox[s] = (min u, min v, max u, max v).
 * pointcloud (O,N,3), poses (T,4,4), trans (S,3), K (3,3) f32; obj/tmpl (S) i32 -> uv (S,N,2) i32, bbox (S,4) i32.
 * ref: project_template_to_image, model/detector.py:209-232 and the bbox of compute_geometric_score :316-318. */
int s6d_project_bbox_f32(const float *pointcloud, const float *poses, const int32_t *obj, const int32_t *tmpl,
                         const float *trans, const float *K, int S, int N, int H, int W, int32_t *uv,
                         int32_t *bbox, void *stream);
/* Several frames in one launch: K (F,3,3) f32, frame (S) i32 = the frame of every proposal (NULL: one K). */
int s6d_project_bbox_frames_f32(const float *pointcloud, const float *poses, const int32_t *obj, const int32_t *tmpl,
                                const float *trans, const float *K, const int32_t *frame, int S, int N, int H, int W,
                                int32_t *uv, int32_t *bbox, void *stream);

/* Geometric structure embedding, fused: out[p,:] = W_d s(idx4[p,0]) + b_d + max_k (W_a s(idx4[p,1+k]) + b_a)
 * with s(x) the interleaved sinusoidal embedding [sin(x w_i), cos(x w_i)] (w = div_term (C/2)).
 * idx4 (NP,4) f32 = [d_idx, a_idx_0..2] per point pair (NP = B*N*N); W_d, W_a (C,C) f32 row-major (out,in);
 * out (NP,C) f32.  C = 256, K = 3.  bf16 matrix cores with a 3-term hi/lo split (fp32-class accuracy).
 * ref: GeometricStructureEmbedding.forward, Pose_Estimation_Model/model/transformer.py:334-349 and
 * SinusoidalPositionalEmbedding.forward :263-281. */
int s6d_geo_embedding_f32(const float *idx4, long NP, const float *Wd, const float *bd, const float *Wa,
                          const float *ba, const float *div_term, int C, int K, float *out, void *stream);
/* The same arithmetic with the result STORED in IEEE half: out_f16 (NP,C) f16.  The embedding's only reader is the RPE attention
 * core, which streams it twelve times per forward; read with the _e16 entry points below. */
int s6d_geo_embedding_f16(const float *idx4, long NP, const float *Wd, const float *bd, const float *Wa,
                          const float *ba, const float *div_term, int C, int K, void *out_f16, void *stream);
/* The same with the two weight matrices PRE-SPLIT into their bf16 hi / lo parts: Wd_hilo, Wa_hilo = [hi (C,C) | lo (C,C)] bf16 as
 * s6d_linear_split_weight_f32 makes them (once per weight version; 16-byte aligned) -- the in-kernel split of the other two entry
 * points is half of their vector instructions.  out: (NP,C) f32, or f16 when out_f16 != 0.  Same values bit for bit. */
int s6d_geo_embedding_split(const float *idx4, long NP, const void *Wd_hilo, const float *bd, const void *Wa_hilo,
                            const float *ba, const float *div_term, int C, int K, void *out, int out_f16, void *stream);
/* Which kernel serves s6d_geo_embedding_split (process-wide, like s6d_set_gemm_wave_tile): 1 = the two-phase kernel of rounds 3-5
 * (default: 2 % ahead in the benched step), 2 = geo_embed2_kernel (round 6: sinusoid fragments built in registers, only the weight
 * slices in LDS by LDS-DMA, one barrier per k-step, no LDS bank conflicts; profiles/r06_geo_embed.md).  Same bits either way.
 * Other values: S6D_EINVAL. */
int s6d_set_geo_embed_form(int form);

/* Fused fp32 Linear of the point transformer:  y = LN( res + act( x W^T + b ) )  with every stage optional.
 * x (M,K) f32 row stride ldx; W given as its bf16 hi / lo parts (N,K) each, made once per weight version by
 * s6d_linear_split_weight_f32 (w = hi + lo up to 2^-17 relative); bias (N) or NULL; act 0 = none, 1 = ReLU; res (M,N) f32
 * row stride ldr or NULL; gamma / beta (N) = LayerNorm over the N outputs with eps, or both NULL (N == 256 when given);
 * y (M,N) f32 row stride ldy.  K % 32 == 0, N % 256 == 0, strides % 4 == 0, 16-byte aligned pointers.  fp32-class products on the
 * bf16 matrix cores (3-term split), fp32 epilogue.
 * ref: the nn.Linear / ReLU / residual / nn.LayerNorm statements of Pose_Estimation_Model/model/transformer.py:93-148 (proj_q/k/v),
 * :182-197 (AttentionOutput), :200-224 and :572-590 (linear + residual + norm), coarse_/fine_point_matching.py in_proj / out_proj. */
int s6d_linear_f32(const float *x, long ldx, int M, int K, const void *w_hi, const void *w_lo, const float *bias, int N, int act,
                   const float *res, long ldr, const float *gamma, const float *beta, float eps, float *y, long ldy,
                   void *stream);

/* The post-attention chain of a PEM transformer layer in ONE launch (csrc/s6d_pchain.hip):
 *   h = LN1(x + a W1^T + b1);  y = LN2(h + relu(h We^T + be) Ws^T + bs)
 * = AttentionLayer / RPEAttentionLayer / LinearAttentionLayer's `norm(linear(attention) + x)` followed by AttentionOutput
 * (Pose_Estimation_Model/model/transformer.py:182-197, :200-224, :409-438, :567-608).  a, x, y (M,256) f32 with row strides lda / ldx /
 * ldy (multiples of 4, 16-byte aligned bases); W1 (256,256), We (512,256), Ws (256,512) as the bf16 hi / lo parts made by
 * s6d_linear_split_weight_f32 AND re-arranged by s6d_linear_fragment_weight (each wave loads its matrix-instruction operands
 * straight from global memory: W never passes through LDS); biases and LayerNorm parameters f32.  The arithmetic is s6d_linear_f32's,
 * operation for operation: the result equals three s6d_linear_f32 launches bit for bit.
 * s6d_linear_fragment_weight: w (N,K) bf16 row-major (N % 32 == 0, K % 16 == 0) -> out, the same elements in fragment order: for row
 * tile t (32 rows) and k step s (16 wide) the 64 lanes' 8-element fragments back to back -- element (32 t + (lane & 31),
 * 16 s + 8 (lane >> 5) + e) at ((t K / 16 + s) 64 + lane) 8 + e. */
int s6d_linear_fragment_weight(const void *w, int N, int K, void *out, void *stream);
int s6d_attn_output_chain_f32(const float *a, long lda, const float *x, long ldx, int M, const void *w1_hi, const void *w1_lo,
                              const float *b1, const float *gamma1, const float *beta1, float eps1, const void *we_hi, const void *we_lo,
                              const float *be, const void *ws_hi, const void *ws_lo, const float *bs, const float *gamma2,
                              const float *beta2, float eps2, float *y, long ldy, void *stream);
/* The TOKEN side of a SAM TwoWayAttentionBlock (csrc/s6d_samtok.hip; segment_anything/modeling/transformer.py:109-186 steps 1-3 for
 * the sparse tokens, :189-240 Attention) as two launches around the token->image attention core (s6d_samdec_tok2img_raw_bf16):
 *   pre : x = queries (+ pe when add_pe);  a = self_attn(q = x, k = x, v = queries);  q1 = norm1(add_pe ? queries + a : a);
 *         qp = cross_attn_token_to_image.q_proj(q1 + pe)
 *   post: q2 = norm2(q1 + cross_attn_token_to_image.out_proj(att));  q3 = norm3(q2 + mlp(q2));
 *         kt = cross_attn_image_to_token.k_proj(q3 + pe);  vt = cross_attn_image_to_token.v_proj(q3)
 * queries, pe, q1, q3 (B,T,256) f32; qp, att, kt, vt (B,T,128) f32; T <= 8 tokens per prompt; all 16-byte aligned, contiguous.
 * Every weight is bf16 (N,K) in the FRAGMENT ORDER of s6d_linear_fragment_weight (wq / wk / wv / wo (256,256), wq2 / wk3 / wv3
 * (128,256), wo2 (256,128), w1 (2048,256), w2 (256,2048)); biases and LayerNorm parameters f32.  Arithmetic = the bf16 autocast
 * statement of the reference modules: a Linear multiplies bf16-rounded activations by bf16 weights, accumulates in fp32 and rounds
 * its result to bf16; LayerNorm, residual adds and the softmax are fp32.
 * Optional operands (null = off) fold the small per-head products around the attention cores into the two launches:
 *   pre : wkfold (256,128) = cross_attn_token_to_image.k_proj.weight^T in fragment order -> qfold_out (B,64,256) bf16, row h*8 + t =
 *         bf16(fold_scale * sum_d qp[b,t,16h+d] Wk[16h+d,:]) (zero for t >= T): the query operand of s6d_samdec_tok2img_raw_bf16
 *         (fold_scale = log2(e) / sqrt(16)); qp_out may then be null;
 *   post: y (B,64,256) f32 = that kernel's result, wvfold (128,256) = v_proj.weight in fragment order, bvfold (128):
 *         att[b,t,16h+d] = sum_c y[b,h*8+t,c] Wv[16h+d,c] + bv is formed in the kernel (att may then be null);
 *         wqfold (256,128) = cross_attn_image_to_token.q_proj.weight^T, bqfold (128), wofold (256,128) = its out_proj.weight, both in
 *         fragment order -> the operands of s6d_samdec_img2tok[_raw]_bf16: kexp_out (B,64,128) bf16 block-diagonal kt / 4 (the caller
 *         zero-fills it), k256_out (B,64,256) bf16 = kexp Wq, cb_out (B,64) f32 = kexp . bq, vpt_out (B,256,64) bf16 =
 *         [sum_d vt[b,t,16h+d] Wo[n,16h+d]] at [b,n,h*8+t]; slots t >= T are zero; kt_out / vt_out may then be null. */
int s6d_samdec_tokens_pre_bf16(const float *queries, const float *pe, int B, int T, int add_pe, const void *wq, const float *bq,
                               const void *wk, const float *bk, const void *wv, const float *bv, const void *wo, const float *bo,
                               const float *gamma1, const float *beta1, float eps1, const void *wq2, const float *bq2, float *q1_out,
                               float *qp_out, const void *wkfold, float fold_scale, void *qfold_out, void *stream);
int s6d_samdec_tokens_post_bf16(const float *q1, const float *att, const float *pe, int B, int T, const void *wo2, const float *bo2,
                                const float *gamma2, const float *beta2, float eps2, const void *w1, const float *b1, const void *w2,
                                const float *b2, const float *gamma3, const float *beta3, float eps3, const void *wk3, const float *bk3,
                                const void *wv3, const float *bv3, float *q3_out, float *kt_out, float *vt_out, const float *y,
                                const void *wvfold, const float *bvfold, const void *wqfold, const float *bqfold, const void *wofold,
                                void *kexp_out, void *k256_out, float *cb_out, void *vpt_out, void *stream);
int s6d_linear_split_weight_f32(const float *w, long n, void *hi, void *lo, void *stream);

/* Soft-assignment head of compute_fine_Rt (Pose_Estimation_Model/utils/model_utils.py:262-270), fused.
 * atten (B,M1,M2) f32 similarity / temp (row 0 / col 0 = background token), pts2 (B,M2-1,3) f32 ->
 *   w1   (B,M1-1)   1 if the row's arg-max over softmax(dim=2)*softmax(dim=1) is not the background column
 *   wsum (B,M1-1)   row sums of the masked assignment a_ij = p_ij w1_i w2_j   (i,j >= 1)
 *   pred (B,M1-1,3) (a_i / (wsum_i + 1e-6)) @ pts2
 * |atten| must stay below ~80 (it is cosine/temp = +-10): exp() is taken without a max shift.  M2 <= 2112. */
int s6d_fine_assign_f32(const float *atten, const float *pts2, int B, int M1, int M2, void *workspace, float *pred,
                        float *wsum, float *w1, void *stream);
long s6d_fine_assign_workspace_bytes(int B, int M1, int M2);

/* Fine point matching, similarity + soft assignment fused: the (B, M1, M2) similarity matrix is never written.
 * f1 (B,M1,C), f2 (B,M2,C) f32 = out_proj features INCLUDING the background token at row 0 (un-normalised);
 * pts2 (B,M2-1,3) f32; inv_temp = 1 / temp.  -> pred (B,M1-1,3), wsum (B,M1-1), w1 (B,M1-1) f32 as s6d_fine_assign_f32.
 * C = 256.  Tiles of the matrix are recomputed on the bf16 matrix cores with a hi/lo split of the normalised rows
 * (fp32-class similarities) in three sweeps: row sums, column sums + column labels, row labels + assignment.
 * ref: compute_feature_similarity, utils/model_utils.py:114-136 (called at model/fine_point_matching.py:75-80) +
 * compute_fine_Rt, utils/model_utils.py:250-270. */
int s6d_fine_match_f32(const float *f1, const float *f2, const float *pts2, int B, int M1, int M2, int C, float inv_temp,
                       void *workspace, float *pred, float *wsum, float *w1, void *stream);
long s6d_fine_match_workspace_bytes(int B, int M1, int M2);

/* Plain multi-head self-attention over a token sequence (no positional bias): qkv (B,N,3,nh,hd) bf16 ->
 * out (B,N,nh*hd) bf16, softmax(scale q.k) v; all N key slots LDS-resident (N <= ~500 at hd 64).  hd in {64, 80}.
 * ref: the timm ViT attention used by Pose_Estimation_Model/model/feature_extraction.py:17-35 (N = 197, hd = 64). */
int s6d_seq_attention_bf16(const void *qkv, int B, int N, int num_heads, int head_dim, float scale, void *out,
                           void *stream);

/* PositionalEncoding branch, fused: for every point, gather its `ns` ball-query neighbours, build the 6-channel
 * input [nbr - centre ; nbr], run the 3-layer SharedMLP (BatchNorm folded: W0 (32,6), W1 (64,32), W2 (128,64),
 * row-major (out,in), + biases) with ReLU, and max over the neighbours.
 * pts (B,N,3) f32, idx (B,N,ns) i32 (s6d_ball_query_f32 output) -> out (B,N,128) f32.  Layers 1 and 2 run on the bf16 matrix
 * cores with the 3-term split of s6d_linear_f32 (fp32 accumulation, ~2^-17 relative per product; layer 0 in fp32).
 * ref: PositionalEncoding.forward, Pose_Estimation_Model/model/fine_point_matching.py:101-117 (one scale),
 * QueryAndGroup pointnet2_utils.py:331-356, SharedMLP pytorch_utils.py:25-50. */
int s6d_pe_group_mlp_f32(const float *pts, const int32_t *idx, int B, int N, int ns, const float *W0, const float *b0,
                         const float *W1, const float *b1, const float *W2, const float *b2, float *out,
                         void *stream);

/* Sam.preprocess: out[b,c,y,x] = (in[b,c,y,x] - mean[c]) / std[c] for y < h, x < w, else 0 (zero pad to S x S),
 * written in the encoder's dtype (out_bf16 != 0: bf16, else f32).  in (B,3,h,w) f32; mean/std: 3 floats in HOST
 * memory.  ref: segment_anything/modeling/sam.py:164-174. */
int s6d_sam_preprocess_f32(const float *in, int B, int h, int w, int S, const float *mean3_host,
                           const float *std3_host, int out_bf16, void *out, void *stream);

/* flags[b] = 1 if row b of x (B rows of n floats, contiguous, n % 4 == 0, x 16-byte aligned) holds an inf or a NaN, else 0.
 * The range guard of the IEEE-half PEM feature extractor (one read of its fp32 up-projection instead of a library reduction whose
 * SUM could itself overflow); ref: ViT_AE.forward, Pose_Estimation_Model/model/feature_extraction.py:98-117 (the reference runs
 * fp32 and has no guard). */
int s6d_nonfinite_rows_f32(const float *x, int B, long n, int32_t *flags, void *stream);

/* The A operand of a 3x3, padding-1 convolution run as one GEMM over K = 9 C: in (B,H,W,C) of two-byte elements (bf16 or f16,
 * copied bit for bit) -> out (B,H,W,9C), out[b,y,x,(3 dy + dx) C + c] = in[b, y + dy - 1, x + dx - 1, c], 0 outside the map.
 * C % 8 == 0.  ref: the neck's second Conv2d, segment_anything/modeling/image_encoder.py:91-97 (weight flattened in
 * (dy, dx, ci) order by the caller). */
int s6d_im2col3x3_b16(const void *in, int B, int H, int W, int C, void *out, void *stream);

/* The A operand of a kernel = stride = p convolution (PatchEmbed): in (B,Cin,H,W) two-byte elements -> out (B,H/p,W/p,Cin p p),
 * a patch's elements in (c, dy, dx) order = Conv2d.weight.flatten(1)'s.  p % 8 == 0, H % p == W % p == 0.
 * ref: PatchEmbed.forward, segment_anything/modeling/image_encoder.py:375-395. */
int s6d_patchify_b16(const void *in, int B, int Cin, int H, int W, int p, void *out, void *stream);

/* Features of chosen pixels of the x4-upsampled feature map without materialising it.
 * up (B, G*G, P*P, C) f32 = output_upscaling(tokens) before the pixel shuffle (G = 14 patches, P = 4, C = 256);
 * choose (B,n) int64 pixel ids in the H x W image -> out (B,n,C) f32, bilinear (align_corners = False).
 * ref: ViT_AE.forward, Pose_Estimation_Model/model/feature_extraction.py:111-114 + get_chosen_pixel_feats,
 * utils/model_utils.py:69-81. */
int s6d_upsample_gather_f32(const float *up, const int64_t *choose, int B, int n, int G, int P, int C, int H, int W,
                            float *out, void *stream);

/* Plain multi-head attention rows (no positional term): q (B,N,C), k, v (B,M,C) f32 projected, heads "(h c)" ->
 * out (B,N,C).  C = 256, heads = 4.  ref: MultiHeadAttention.forward, Pose_Estimation_Model/model/transformer.py:93-148
 * (the cross-attention of GeometricTransformer; no masks / factors on the inference path). */
int s6d_mha_f32(const float *q, const float *k, const float *v, int B, int N, int M, int C, int heads, float scale,
                float *out, void *stream);
int s6d_mha_strided_f32(const float *q, long ldq, const float *k, long ldk, const float *v, long ldv, int B, int N, int M, int C,
                        int heads, float scale, float *out, void *stream);   /* row strides as s6d_rpe_attention_strided_f32 */

/* Focused linear attention feature map: t = (relu(x) + 1e-6) * inv_scale; y = t^p / |t^p| * |t| per row.
 * x, y (rows,256) f32; inv_scale (256) = 1 / softplus(scale).  ref: LinearAttention.forward,
 * Pose_Estimation_Model/model/transformer.py:536-547. */
int s6d_linear_attn_focus_f32(const float *x, const float *inv_scale, long rows, int C, int power, float *y,
                              void *stream);

/* The whole focused linear attention behind the projections (LinearAttention.forward, transformer.py:536-564, kv-first branch):
 * q_proj (B,I,C) the raw proj_q output (the focus map is applied inside), k_focused (B,J,C) rows with stride ldk, v (B,J,C) rows with
 * stride ldv, inv_scale (C) = 1 / softplus(scale) -> out (B,I,C) = merge_heads((q kv) / (q . sum_j k_j + 1e-6)).  C = 256 (4 heads of
 * 64).  kv_ws: s6d_linear_attention_workspace_floats(B) floats of scratch. */
long s6d_linear_attention_workspace_floats(int B);
int s6d_linear_attention_f32(const float *q_proj, const float *inv_scale, int power, const float *k_focused, long ldk, const float *v,
                             long ldv, int B, int I, int J, int C, float *kv_ws, float *out, void *stream);

/* Proposal crops for the descriptor model, fused: normalise . mask . crop . nearest resize . zero pad . nearest resize.
 * image (H,W,3) u8, masks (P,H,W) f32 {0,1}, params: P records of 12 int32 / float32 words
 *   {x1, y1, h, w, h1, w1, top, left, S2, float inv1, float inv2, 0}
 * (crop origin and size, size after the first resize, padding in front, padded square side, float(1/scale) of the two
 * resizes -- the host mirrors the reference loop's double-precision size arithmetic).  out_rgb (P,3,T,T) and/or
 * out_mask (P,T,T) f32, either may be NULL.  mean / std: 3 floats in HOST memory.  Bit-exact with the reference.
 * ref: CustomDINOv2.process_rgb_proposals / process_masks_proposals, Instance_Segmentation_Model/model/dinov2.py:131-144,
 * 178-189; CropResizePad.__call__, utils/bbox_utils.py:98-126 (+ torchvision ToTensor / Normalize, dinov2.py:115-120). */
int s6d_crop_resize_pad_f32(const unsigned char *image, const float *masks, const void *params, int P, int H, int W,
                            int T, const float *mean3_host, const float *std3_host, float *out_rgb, float *out_mask,
                            void *stream);

/* SAM mask decoder, image -> token cross attention of a two-way block, fused with out_proj, the residual add and
 * norm4:  out = LayerNorm(resid + softmax(q k^T / sqrt(16)) (v W_o^T) + b_o)  for every image token, 8 heads x 16, at
 * most 8 prompt tokens.  q (1|B,N,128) bf16 projected image tokens, row stride q_ld elements (+ q_add (N,128) bf16, e.g.
 * W_q pe, or NULL);
 * kexp (B,64,128) bf16: row h*8+t = k_t / 4 in head h's 16 columns, zeros elsewhere; vpt (B,256,64) bf16:
 * vpt[n][h*8+t] = sum_d v_t[h*16+d] W_o[n][h*16+d]; resid (1|B,N,256) bf16; out_bias, ln_w, ln_b (256) f32 ->
 * out (B,N,256) bf16.  q_shared / resid_shared: that operand has no batch dimension (layer 0).
 * ref: TwoWayAttentionBlock.forward step (4), segment_anything/modeling/transformer.py:174-180; Attention.forward
 * :222-240. */
int s6d_samdec_img2tok_bf16(const void *q, const void *q_add, const void *kexp, const void *vpt, const void *resid,
                            const float *out_bias, const float *ln_w, const float *ln_b, float ln_eps, int B, int N,
                            int n_tok, int q_ld, int q_shared, int resid_shared, void *out, void *stream);

/* The same block with the q projection of the image tokens folded into the expanded keys: x (1|B,N,256) bf16 RAW image tokens with row
 * stride x_ld (x_shared: no batch dimension), pe (N,256) bf16 added to x for the scores (bf16 rounding) or NULL,
 * kexp256 (B,64,256) bf16 = kexp W_q (kexp as above, W_q (128,256) the q_proj weight), cbias (B,64) f32 = kexp . b_q (the bias of
 * the projection differs between the slots a token's softmax runs over); the other operands as s6d_samdec_img2tok_bf16.  The
 * (B,N,128) projected queries are never written or read.  ref: transformer.py:173-180, Attention.forward :222-240. */
int s6d_samdec_img2tok_raw_bf16(const void *x, const void *pe, const void *kexp256, const float *cbias, const void *vpt,
                                const void *resid, const float *out_bias, const float *ln_w, const float *ln_b, float ln_eps, int B,
                                int N, int n_tok, int x_ld, int x_shared, int resid_shared, void *out, void *stream);

/* SAM mask decoder, token -> image cross attention before out_proj: softmax(q_t k^T / sqrt(16)) v over the N image
 * tokens, 8 heads x 16.  qt (B,8,128) f32 projected prompt tokens (pad unused rows with anything finite; their output
 * rows are meaningless); k and v are the 128-wide column ranges [k_off, k_off+128), [v_off, v_off+128) of a bf16 tensor
 * kv (1|B,N,ld) (kv_shared: no batch dimension); k_pe (N,128) bf16 is added to k (bf16 rounding) or NULL;
 * scale = 1/sqrt(16) -> out (B,8,128) f32.
 * ref: TwoWayAttentionBlock.forward steps (2), segment_anything/modeling/transformer.py:160-165; TwoWayTransformer.forward
 * :98-103; Attention.forward :222-240. */
int s6d_samdec_tok2img_f32(const float *qt, const void *kv, int ld, int k_off, int v_off, int kv_shared, const void *k_pe,
                           int B, int N, float scale, float *out, void *stream);

/* The same attention on the RAW image tokens, the k / v projections folded into the queries (no k / v tensor over the B x N image
 * tokens): qp (B,64,256) bf16, row j = head * 8 + slot = scale * log2(e) * W_k[head]^T q[slot, head] (zero rows for unused slots);
 * x (1|B,N,256) bf16 with row stride ld (x_shared: no batch dimension); pe (N,256) bf16 added to x for the scores (bf16 rounding) or
 * NULL -> y (B,64,256) f32, y[j] = sum_n softmax_n(qp[j] . (x_n + pe_n)) x_n.  The caller finishes with out[slot, head] =
 * W_v[head] y[head * 8 + slot] + b_v[head].  N % 64 == 0.
 * ref: as s6d_samdec_tok2img_f32 (transformer.py:160-165, :98-103, :222-240). */
int s6d_samdec_tok2img_raw_bf16(const void *qp, const void *x, int ld, int x_shared, const void *pe, int B, int N, float *y,
                                void *stream);

/* SAM mask decoder, output head after the first transposed conv: LayerNorm2d + GELU, second 2x2/2 transposed conv,
 * GELU, and the hypernetwork product, per output pixel.  y0 (B,h*w,4*64) bf16, row stride y_ld elements = first
 * transposed conv as a GEMM with columns ordered (dy,dx,c); w2t (128,64) bf16: row (dy2*2+dx2)*32+ch; hyper (B,M,32) f32, M <= 4 ->
 * masks (B,M,4h,4w) f32 logits.  GELU is the exact (erf) form, erf to 1.5e-7.
 * ref: MaskDecoder.predict_masks, segment_anything/modeling/mask_decoder.py:126-139 (output_upscaling :53-59). */
int s6d_samdec_upscale_heads_bf16(const void *y0, const float *ln_w, const float *ln_b, float ln_eps, const void *w2t,
                                  const float *b2, const float *hyper, int B, int M, int h, int w, int y_ld,
                                  float *masks, void *stream);

/* Mask post-processing of SAM's automatic mask generator, fused: bilinear upscaling of the (Bm,n,n) f32 mask logits to
 * the padded square (img_size), crop to (in_h,in_w), bilinear to the frame (H,W) -- evaluated per frame pixel straight
 * from the low-resolution logits -- then masks (Bm,H,W) u8 = logit > mask_threshold and stats (Bm,6) i32 =
 * {#(logit > thr + offset), #(logit > thr - offset), x_min, y_min, x_max, y_max of the mask (W, H, -1, -1 if empty)}.
 * Arithmetic = ATen's CPU upsample_bilinear2d (align_corners = False), operation for operation: bit-identical masks.
 * ref: Sam.postprocess_masks, segment_anything/modeling/sam.py:133-162; calculate_stability_score / batched_mask_to_box,
 * segment_anything/utils/amg.py:156-176, 303-346; SamAutomaticMaskGenerator._process_batch :281-312. */
int s6d_sam_mask_post_f32(const float *low_res, int Bm, int n, int img_size, int in_h, int in_w, int H, int W,
                          float mask_threshold, float stability_offset, unsigned char *masks, int32_t *stats,
                          void *stream);
/* The same on a channel slice of the decoder's output, read in place: low_res (B,ch_total,n,n) f32, masks of channels
 * [ch_first, ch_first + ch_count) -> masks (B*ch_count,H,W) u8 (0 / 1: also valid as a bool tensor), stats (B*ch_count,6).
 * `masks[:, 1:]` of MaskDecoder.forward(multimask_output=True), mask_decoder.py:99-104, without the contiguous copy. */
int s6d_sam_mask_post_sel_f32(const float *low_res, int B, int ch_total, int ch_first, int ch_count, int n, int img_size,
                              int in_h, int in_w, int H, int W, float mask_threshold, float stability_offset,
                              unsigned char *masks, int32_t *stats, void *stream);

/* Box NMS with torchvision.ops.nms semantics (IoU on float32 XYXY boxes, area = (x2-x1)(y2-y1), suppress if IoU >
 * threshold, boxes visited in the given score order).  boxes (N,4) f32, order (N) i64 = indices by decreasing score ->
 * keep_sorted (N) u8: keep_sorted[i] says whether the i-th box IN SCORE ORDER survives.  workspace:
 * s6d_nms_workspace_bytes(N) bytes; N <= 16384.
 * ref: batched_nms call sites, segment_anything/automatic_mask_generator.py:251-257 (one category: plain nms);
 * torchvision (un-vendored dependency, ISM environment.yaml) ops/nms: restated from its published algorithm. */
int s6d_nms_f32(const float *boxes, const int64_t *order, int N, float iou_threshold, void *workspace,
                unsigned char *keep_sorted, void *stream);
long s6d_nms_workspace_bytes(int N);

#ifdef __cplusplus
}
#endif
#endif /* SAM6D_HIP_H */
