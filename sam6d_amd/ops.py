"""Thin torch-tensor front-end of the C ABI (include/sam6d_hip.h).

torch is plumbing here: device memory, the current HIP stream and dtype/shape checks.
Every function enqueues hand-written gfx950 kernels on torch's current stream and
returns freshly allocated outputs; none of them has a CPU implementation.
"""
import ctypes
import os

import torch

from . import _lib
from . import policy

_vp = ctypes.c_void_p


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return _vp(t.data_ptr())


def _chk(t, dtype, name, ndim=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # same wording as the reference's CHECK_CUDA
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = {torch.float32: "a float", torch.int32: "an int", torch.bfloat16: "a bfloat16"}.get(dtype, str(dtype))
        raise RuntimeError(f"{name} must be {kind} tensor")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name} must have {ndim} dimensions")


def _call(fn_name, *args):
    fn = getattr(_lib.lib(), fn_name)
    fn.restype = ctypes.c_int
    _lib.check(fn(*args), fn_name)


# ------------------------------------------------------------------ PointNet++ ops
def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32   [pointnet2._ext.furthest_point_sampling]"""
    _chk(points, torch.float32, "points", 3)
    B, N, _ = points.shape
    out = torch.empty(B, nsamples, dtype=torch.int32, device=points.device)
    tmp = torch.empty(B, N, dtype=torch.float32, device=points.device) if N > 4096 else None
    _call("s6d_fps_f32", _ptr(points), B, N, int(nsamples), _ptr(tmp) if tmp is not None else _vp(0), _ptr(out),
          _stream())
    return out


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M)   [pointnet2._ext.gather_points]"""
    _chk(points, torch.float32, "points", 3)
    _chk(idx, torch.int32, "idx", 2)
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.empty(B, C, M, dtype=torch.float32, device=points.device)
    _call("s6d_gather_points_f32", _ptr(points), _ptr(idx), B, C, N, M, _ptr(out), _stream())
    return out


def gather_rows(src, idx):
    """(B,N,C) f32, (B,M) i32 -> (B,M,C)."""
    _chk(src, torch.float32, "src", 3)
    _chk(idx, torch.int32, "idx", 2)
    B, N, C = src.shape
    M = idx.shape[1]
    out = torch.empty(B, M, C, dtype=torch.float32, device=src.device)
    _call("s6d_gather_rows_f32", _ptr(src), _ptr(idx), B, N, C, M, _ptr(out), _stream())
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,M,3), (B,N,3) -> (B,M,nsample) i32   [pointnet2._ext.ball_query]"""
    _chk(new_xyz, torch.float32, "new_xyz", 3)
    _chk(xyz, torch.float32, "xyz", 3)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    out = torch.empty(B, M, int(nsample), dtype=torch.int32, device=xyz.device)
    _call("s6d_ball_query_f32", _ptr(new_xyz), _ptr(xyz), B, N, M, ctypes.c_float(radius), int(nsample), _ptr(out),
          _stream())
    return out


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S)   [pointnet2._ext.group_points]"""
    _chk(points, torch.float32, "points", 3)
    _chk(idx, torch.int32, "idx", 3)
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.empty(B, C, M, S, dtype=torch.float32, device=points.device)
    _call("s6d_group_points_f32", _ptr(points), _ptr(idx), B, C, N, M, S, _ptr(out), _stream())
    return out


# ------------------------------------------------------------------ PEM pose solvers
def rot_from_h(H):
    """(...,3,3) f32 cross-covariances -> proper rotations (Procrustes)."""
    _chk(H, torch.float32, "H")
    n = H.numel() // 9
    out = torch.empty_like(H)
    _call("s6d_rot_from_h_f32", _ptr(H), n, _ptr(out), _stream())
    return out


def weighted_procrustes(src, ref, weights, weight_thresh=0.0, eps=1e-5):
    """src, ref (B,N,3), weights (B,N) f32 -> R (B,3,3), t (B,3): ref ~ src R^T + t (batch-size independent, fixed order)."""
    _chk(src, torch.float32, "src", 3)
    _chk(ref, torch.float32, "ref", 3)
    _chk(weights, torch.float32, "weights", 2)
    B, N, _ = src.shape
    if ref.shape != src.shape or weights.shape != (B, N):
        raise ValueError(f"weighted_procrustes: src {tuple(src.shape)}, ref {tuple(ref.shape)}, weights {tuple(weights.shape)}")
    R = torch.empty(B, 3, 3, dtype=torch.float32, device=src.device)
    t = torch.empty(B, 3, dtype=torch.float32, device=src.device)
    _call("s6d_weighted_procrustes_f32", _ptr(src), _ptr(ref), _ptr(weights), B, N, ctypes.c_float(weight_thresh), ctypes.c_float(eps), _ptr(R), _ptr(t),
          _stream())
    return R, t


def pose_hypotheses(pts1, pts2, pair):
    """pts1 (B,N1,3), pts2 (B,N2,3) f32, pair (B,3*n) i32 -> R (B,n,3,3), t (B,n,3), dis (B,n)."""
    _chk(pts1, torch.float32, "pts1", 3)
    _chk(pts2, torch.float32, "pts2", 3)
    _chk(pair, torch.int32, "pair", 2)
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    n = pair.shape[1] // 3
    R = torch.empty(B, n, 3, 3, dtype=torch.float32, device=pts1.device)
    t = torch.empty(B, n, 3, dtype=torch.float32, device=pts1.device)
    dis = torch.empty(B, n, dtype=torch.float32, device=pts1.device)
    _call("s6d_pose_hypotheses_f32", _ptr(pts1), _ptr(pts2), _ptr(pair), B, N1, N2, n, _ptr(R), _ptr(t), _ptr(dis),
          _stream())
    return R, t, dis


def coarse_sample(atten, rand_u):
    """atten (B,M1,M2) f32, rand_u (B,n_u) f32 -> pair (B,n_u) i32 (flat bin index), w1 (B,M1-1) f32."""
    _chk(atten, torch.float32, "atten", 3)
    _chk(rand_u, torch.float32, "rand_u", 2)
    B, M1, M2 = atten.shape
    pair = torch.empty(B, rand_u.shape[1], dtype=torch.int32, device=atten.device)
    w1 = torch.empty(B, M1 - 1, dtype=torch.float32, device=atten.device)
    _call("s6d_coarse_sample_f32", _ptr(atten), _ptr(rand_u), B, M1, M2, int(rand_u.shape[1]), _ptr(pair), _ptr(w1), _stream())
    return pair, w1


def smallest_k(dis, Rs, ts, k):
    """dis (B,n), Rs (B,n,3,3), ts (B,n,3) f32 -> Rk (B,k,3,3), tk (B,k,3), idx (B,k) i32: the k smallest, ascending."""
    _chk(dis, torch.float32, "dis", 2)
    _chk(Rs, torch.float32, "Rs", 4)
    _chk(ts, torch.float32, "ts", 3)
    B, n = dis.shape
    Rk = torch.empty(B, k, 3, 3, dtype=torch.float32, device=dis.device)
    tk = torch.empty(B, k, 3, dtype=torch.float32, device=dis.device)
    idx = torch.empty(B, k, dtype=torch.int32, device=dis.device)
    _call("s6d_smallest_k_f32", _ptr(dis), _ptr(Rs), _ptr(ts), B, n, int(k), _ptr(Rk), _ptr(tk), _ptr(idx), _stream())
    return Rk, tk, idx


def hypothesis_select(dmin, w1, Rk, tk):
    """dmin (B,P,N), w1 (B,N), Rk (B,P,3,3), tk (B,P,3) f32 -> R (B,3,3), t (B,3) of the best-scoring hypothesis."""
    for a, nm, nd in ((dmin, "dmin", 3), (w1, "w1", 2), (Rk, "Rk", 4), (tk, "tk", 3)):
        _chk(a, torch.float32, nm, nd)
    B, P, N = dmin.shape
    R = torch.empty(B, 3, 3, dtype=torch.float32, device=dmin.device)
    t = torch.empty(B, 3, dtype=torch.float32, device=dmin.device)
    _call("s6d_hypothesis_select_f32", _ptr(dmin), _ptr(w1), _ptr(Rk), _ptr(tk), B, P, N, _ptr(R), _ptr(t), _stream())
    return R, t


def min_dist(pts, R, t, model):
    """pts (B,N,3), R (B,P,3,3), t (B,P,3), model (B,Nm,3) -> (B,P,N)."""
    for a, nm, nd in ((pts, "pts", 3), (R, "R", 4), (t, "t", 3), (model, "model", 3)):
        _chk(a, torch.float32, nm, nd)
    B, N, _ = pts.shape
    P, Nm = R.shape[1], model.shape[1]
    out = torch.empty(B, P, N, dtype=torch.float32, device=pts.device)
    _call("s6d_min_dist_f32", _ptr(pts), _ptr(R), _ptr(t), _ptr(model), B, N, P, Nm, _ptr(out), _stream())
    return out


# ------------------------------------------------------------------ PEM point transformer
def _rows3(t, name):
    """(B, N, 256) f32 with unit channel stride whose rows sit at a constant stride (a column block of a wider contiguous (B, N, L)
    tensor qualifies) -> (tensor, row stride); anything else is made contiguous."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3:
        raise RuntimeError(f"{name} must be a 3-d float32 CUDA tensor")
    B, N, C = t.shape
    if t.stride(2) == 1 and t.stride(1) % 4 == 0 and t.stride(1) >= C and (B == 1 or t.stride(0) == N * t.stride(1)) \
            and t.data_ptr() % 16 == 0:
        return t, t.stride(1)
    t = t.contiguous()
    return t, C


def rpe_attention(q, k, v, qt, qb, embed, scale):
    """q,k,v (B,N,256) (column blocks of one projection output are taken as they are); qt (B,4,N,256); qb (B,4,N);
    embed (B,N,N,256) -> (B,N,256), all f32."""
    (q, ldq), (k, ldk), (v, ldv) = _rows3(q, "q"), _rows3(k, "k"), _rows3(v, "v")
    qt, qb = qt.contiguous(), qb.contiguous()
    for a, nm in ((qt, "qt"), (qb, "qb")):
        _chk(a, torch.float32, nm)
    _chk(embed, embed.dtype if embed.dtype == torch.float16 else torch.float32, "embed", 4)
    B, N, C = q.shape
    out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
    _call("s6d_rpe_attention_strided_e16_f32" if embed.dtype == torch.float16 else "s6d_rpe_attention_strided_f32", _ptr(q), ctypes.c_long(ldq), _ptr(k), ctypes.c_long(ldk), _ptr(v), ctypes.c_long(ldv),
          _ptr(qt), _ptr(qb), _ptr(embed), B, N, C, 4, ctypes.c_float(scale), _ptr(out), _stream())
    return out


def rpe_attention_packed(proj, embed, scale, q_off=0, k_off=256, v_off=512, qt_off=768, qb_off=1792):
    """rpe_attention on ONE projection output proj (B,N,ld) f32 holding q | k | v | q~ (4 x 256) | qb (4) as column blocks (W_p folded
    into the projection's weights by the caller) -> (B,N,256) f32."""
    _chk(proj, torch.float32, "proj", 3)
    _chk(embed, embed.dtype if embed.dtype == torch.float16 else torch.float32, "embed", 4)
    B, N, ld = proj.shape
    out = torch.empty(B, N, 256, dtype=torch.float32, device=proj.device)
    _call("s6d_rpe_attention_packed_e16_f32" if embed.dtype == torch.float16 else "s6d_rpe_attention_packed_f32", _ptr(proj), ctypes.c_long(ld), int(q_off), int(k_off), int(v_off), int(qt_off), int(qb_off),
          _ptr(embed), B, N, 256, 4, ctypes.c_float(scale), _ptr(out), _stream())
    return out


# ------------------------------------------------------------------ edges of the path
def sam_preprocess(x, mean, std, img_size, out_dtype=torch.bfloat16):
    """(B,3,h,w) f32 -> (B,3,S,S) normalised + zero padded, in `out_dtype` (bf16 or f32).  [Sam.preprocess]"""
    _chk(x, torch.float32, "x", 4)
    B, C, h, w = x.shape
    if C != 3:
        raise RuntimeError("x must have 3 channels")
    out = torch.empty(B, 3, img_size, img_size, dtype=out_dtype, device=x.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    _call("s6d_sam_preprocess_f32", _ptr(x), B, h, w, int(img_size), m, s, 1 if out_dtype == torch.bfloat16 else 0,
          _ptr(out), _stream())
    return out


def nonfinite_rows(x):
    """x (B, ...) f32 contiguous -> (B,) bool: does instance b hold an inf or a NaN?  One read of x."""
    _chk(x, torch.float32, "x")
    B = x.shape[0]
    n = x.numel() // max(B, 1)
    flags = torch.empty(B, dtype=torch.int32, device=x.device)
    _call("s6d_nonfinite_rows_f32", _ptr(x), B, ctypes.c_long(n), _ptr(flags), _stream())
    return flags.bool()


def im2col3x3(y):
    """(B,H,W,C) bf16 / f16 -> (B,H,W,9C): the nine shifted views of a zero-padded 3x3 neighbourhood in (dy, dx, c) order -- the A
    operand of a padding-1 3x3 convolution as one GEMM.  [the SAM neck's second Conv2d, image_encoder.py:91-97]"""
    if y.dtype not in (torch.bfloat16, torch.float16) or y.dim() != 4 or not y.is_contiguous():
        raise RuntimeError("im2col3x3: contiguous (B,H,W,C) bf16 / f16 expected")
    B, H, W, C = y.shape
    out = torch.empty(B, H, W, 9 * C, dtype=y.dtype, device=y.device)
    _call("s6d_im2col3x3_b16", _ptr(y), B, H, W, C, _ptr(out), _stream())
    return out


def patchify(x, p):
    """(B,Cin,H,W) bf16 / f16 -> (B,H/p,W/p,Cin*p*p) with a patch's elements in (c, dy, dx) order (Conv2d.weight.flatten(1)'s):
    the A operand of a kernel = stride = p convolution.  [PatchEmbed.forward, image_encoder.py:375-395]"""
    if x.dtype not in (torch.bfloat16, torch.float16) or x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError("patchify: contiguous (B,Cin,H,W) bf16 / f16 expected")
    B, Cin, H, W = x.shape
    out = torch.empty(B, H // p, W // p, Cin * p * p, dtype=x.dtype, device=x.device)
    _call("s6d_patchify_b16", _ptr(x), B, Cin, H, W, int(p), _ptr(out), _stream())
    return out


def crop_resize_pad(image_u8, masks, params, target, mean, std, rgb=True, mask=True):
    """Fused proposal crops [CustomDINOv2.process_rgb_proposals / process_masks_proposals].
    image_u8 (H,W,3) uint8, masks (P,H,W) f32, params (P,12) int32 records (sam6d_amd.ism.dinov2.crop_params)
    -> (rgbs (P,3,T,T) f32 or None, masks (P,T,T) f32 or None)."""
    _chk(masks, torch.float32, "masks", 3)
    _chk(params, torch.int32, "params", 2)
    P, H, W = masks.shape
    if rgb:
        _chk(image_u8, torch.uint8, "image", 3)
        if tuple(image_u8.shape) != (H, W, 3):
            raise RuntimeError("crop_resize_pad: image must be (H,W,3) like the masks")
    if tuple(params.shape) != (P, 12):
        raise RuntimeError("crop_resize_pad: params must be (P,12)")
    T = int(target)
    o_rgb = torch.empty(P, 3, T, T, dtype=torch.float32, device=masks.device) if rgb else None
    o_mask = torch.empty(P, T, T, dtype=torch.float32, device=masks.device) if mask else None
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    _call("s6d_crop_resize_pad_f32", _ptr(image_u8) if rgb else _vp(0), _ptr(masks), _ptr(params), P, H, W, T, m, sd,
          _ptr(o_rgb) if rgb else _vp(0), _ptr(o_mask) if mask else _vp(0), _stream())
    return o_rgb, o_mask


def samdec_img2tok(q, q_add, kexp, vpt, resid, out_bias, ln_w, ln_b, eps, n_tok):
    """Image->token cross attention + folded out_proj + residual + LayerNorm of SAM's two-way block.
    q (1|B,N,128) bf16 (last-dim slice of a wider tensor allowed), q_add (N,128) bf16 or None, kexp (B,64,128) bf16,
    vpt (B,256,64) bf16, resid (1|B,N,256) bf16 -> (B,N,256) bf16."""
    B, N = kexp.shape[0], resid.shape[1]
    for t, nm in ((kexp, "kexp"), (vpt, "vpt"), (resid, "resid")):
        _chk(t, torch.bfloat16, nm, 3)
    for t, nm in ((out_bias, "out_bias"), (ln_w, "ln_w"), (ln_b, "ln_b")):
        _chk(t, torch.float32, nm, 1)
    if q.dtype != torch.bfloat16 or q.dim() != 3 or q.shape[-1] != 128 or q.stride(2) != 1 or not q.is_cuda:
        raise RuntimeError("q must be a (.,N,128) bfloat16 CUDA tensor with unit channel stride")
    q_ld = q.stride(1)
    if q.shape[0] > 1 and q.stride(0) != N * q_ld:
        raise RuntimeError("q batch stride must be N * row stride")
    if q_add is not None:
        _chk(q_add, torch.bfloat16, "q_add", 2)
    if tuple(kexp.shape) != (B, 64, 128) or tuple(vpt.shape) != (B, 256, 64) or resid.shape[2] != 256 or q.shape[1] != N:
        raise RuntimeError("samdec_img2tok: shape mismatch")
    out = torch.empty(B, N, 256, dtype=torch.bfloat16, device=kexp.device)
    _call("s6d_samdec_img2tok_bf16", _ptr(q), _ptr(q_add) if q_add is not None else _vp(0), _ptr(kexp), _ptr(vpt),
          _ptr(resid), _ptr(out_bias), _ptr(ln_w), _ptr(ln_b), ctypes.c_float(eps), B, N, int(n_tok), int(q_ld),
          1 if q.shape[0] == 1 and B > 1 else 0, 1 if resid.shape[0] == 1 and B > 1 else 0, _ptr(out), _stream())
    return out


def samdec_img2tok_raw(x, pe, kexp256, cbias, vpt, resid, out_bias, ln_w, ln_b, eps, n_tok):
    """samdec_img2tok with the q projection folded into the expanded keys: x (1|B,N,256) bf16 RAW image tokens, pe (N,256) bf16 or
    None (added to x for the scores), kexp256 (B,64,256) bf16 = kexp_128 @ W_q, cbias (B,64) f32 = kexp_128 @ b_q, vpt (B,256,64) bf16,
    resid (1|B,N,256) bf16 -> (B,N,256) bf16.  The projected queries of the B x N image tokens are never written."""
    B, N = kexp256.shape[0], resid.shape[1]
    for t, nm in ((kexp256, "kexp256"), (vpt, "vpt"), (resid, "resid"), (x, "x")):
        _chk(t, torch.bfloat16, nm, 3)
    for t, nm in ((out_bias, "out_bias"), (ln_w, "ln_w"), (ln_b, "ln_b")):
        _chk(t, torch.float32, nm, 1)
    _chk(cbias, torch.float32, "cbias", 2)
    if pe is not None:
        _chk(pe, torch.bfloat16, "pe", 2)
    if (tuple(kexp256.shape) != (B, 64, 256) or tuple(vpt.shape) != (B, 256, 64) or resid.shape[2] != 256 or tuple(x.shape[1:]) != (N, 256)
            or tuple(cbias.shape) != (B, 64) or x.shape[0] not in (1, B) or resid.shape[0] not in (1, B)):
        raise RuntimeError("samdec_img2tok_raw: shape mismatch")
    out = torch.empty(B, N, 256, dtype=torch.bfloat16, device=kexp256.device)
    _call("s6d_samdec_img2tok_raw_bf16", _ptr(x), _ptr(pe) if pe is not None else _vp(0), _ptr(kexp256), _ptr(cbias), _ptr(vpt),
          _ptr(resid), _ptr(out_bias), _ptr(ln_w), _ptr(ln_b), ctypes.c_float(eps), B, N, int(n_tok), 256,
          1 if x.shape[0] == 1 and B > 1 else 0, 1 if resid.shape[0] == 1 and B > 1 else 0, _ptr(out), _stream())
    return out


def samdec_tok2img(qt, kv, k_off, v_off, k_pe, scale):
    """Token->image attention before out_proj.  qt (B,T<=8,128) f32, kv (1|B,N,ld) bf16 holding k / v at column
    offsets k_off / v_off, k_pe (N,128) bf16 or None -> (B,T,128) f32."""
    _chk(kv, torch.bfloat16, "kv", 3)
    B, T, _ = qt.shape
    if qt.dtype != torch.float32 or not qt.is_cuda or qt.shape[2] != 128 or T > 8 or kv.shape[0] not in (1, B):
        raise RuntimeError("samdec_tok2img: qt must be (B,T<=8,128) float32 CUDA, kv (1|B,N,ld)")
    q8 = torch.zeros(B, 8, 128, dtype=torch.float32, device=qt.device)
    q8[:, :T] = qt
    if k_pe is not None:
        _chk(k_pe, torch.bfloat16, "k_pe", 2)
    out = torch.empty(B, 8, 128, dtype=torch.float32, device=qt.device)
    _call("s6d_samdec_tok2img_f32", _ptr(q8), _ptr(kv), int(kv.shape[2]), int(k_off), int(v_off),
          1 if kv.shape[0] == 1 and B > 1 else 0, _ptr(k_pe) if k_pe is not None else _vp(0), B, int(kv.shape[1]),
          ctypes.c_float(scale), _ptr(out), _stream())
    return out[:, :T]


def samdec_tok2img_raw(qt, x, pe, wk, wv, bv, scale):
    """Token->image attention before out_proj on the RAW image tokens: the k / v projections are folded into the queries, no
    projected k / v tensor over the image tokens exists.  qt (B,T<=8,128) f32 projected prompt tokens, x (1|B,N,256) bf16 image
    tokens, pe (N,256) bf16 or None (added to x for the scores), wk / wv (128,256), bv (128): the k_proj / v_proj parameters ->
    (B,T,128) f32 = softmax(q (W_k (x + pe))^T * scale) (W_v x + b_v), heads 8 x 16."""
    _chk(x, torch.bfloat16, "x", 3)
    B, T, _ = qt.shape
    if qt.dtype != torch.float32 or not qt.is_cuda or qt.shape[2] != 128 or T > 8 or x.shape[0] not in (1, B) or x.shape[2] != 256:
        raise RuntimeError("samdec_tok2img_raw: qt must be (B,T<=8,128) float32 CUDA, x (1|B,N,256) bfloat16")
    if pe is not None:
        _chk(pe, torch.bfloat16, "pe", 2)
    H, hd = 8, 16
    with torch.autocast(device_type="cuda", enabled=False):          # the two small folds stay float32 inside an autocast region
        # q'[b, h, t, :] = scale * log2(e) * W_k[h]^T q[b, t, h]
        qp = torch.zeros(B, H, 8, 256, dtype=torch.float32, device=qt.device)
        qp[:, :, :T] = torch.einsum("bthd,hdc->bhtc", qt.view(B, T, H, hd) * (scale * 1.4426950408889634), wk.float().view(H, hd, 256))
        qp = qp.to(torch.bfloat16).view(B, 64, 256)
        y = torch.empty(B, 64, 256, dtype=torch.float32, device=qt.device)
        _call("s6d_samdec_tok2img_raw_bf16", _ptr(qp), _ptr(x), int(x.stride(1)), 1 if x.shape[0] == 1 and B > 1 else 0,
              _ptr(pe) if pe is not None else _vp(0), B, int(x.shape[1]), _ptr(y), _stream())
        out = torch.einsum("bhtc,hdc->bthd", y.view(B, H, 8, 256)[:, :, :T], wv.float().view(H, hd, 256)) + bv.float().view(1, 1, H, hd)
    return out.reshape(B, T, 128)


def samdec_tokens_pre(queries, pe, add_pe, lin_q, lin_k, lin_v, lin_o, norm1, lin_q2, fold=None):
    """Self-attention + norm1 + token->image query projection of a TwoWayAttentionBlock's sparse tokens in one launch
    (s6d_samdec_tokens_pre_bf16).  queries, pe (B,T<=8,256) f32; lin_* = (fragment-ordered bf16 weight, f32 bias); norm1 =
    (gamma, beta, eps) -> (q1 (B,T,256) f32, qp (B,T,128) f32).
    fold = (wkfold, scale): the attention core's k projection folded into the queries inside the kernel -> (q1, qfold (B,64,256)
    bf16) instead: the query operand of samdec_tok2img_raw_core."""
    _chk(queries, torch.float32, "queries", 3)
    _chk(pe, torch.float32, "pe", 3)
    B, T, C = queries.shape
    if C != 256 or T > 8 or pe.shape != queries.shape:
        raise ValueError("samdec_tokens_pre: queries / pe must be (B, T <= 8, 256)")
    queries, pe = queries.contiguous(), pe.contiguous()
    q1 = torch.empty(B, T, 256, dtype=torch.float32, device=queries.device)
    if fold is None:
        out2 = torch.empty(B, T, 128, dtype=torch.float32, device=queries.device)
        qp_ptr, wk_ptr, sc, qf_ptr = _ptr(out2), _vp(0), 0.0, _vp(0)
    else:
        out2 = torch.empty(B, 64, 256, dtype=torch.bfloat16, device=queries.device)
        qp_ptr, wk_ptr, sc, qf_ptr = _vp(0), _ptr(fold[0]), float(fold[1]), _ptr(out2)
    _call("s6d_samdec_tokens_pre_bf16", _ptr(queries), _ptr(pe), B, T, 1 if add_pe else 0, _ptr(lin_q[0]), _ptr(lin_q[1]), _ptr(lin_k[0]),
          _ptr(lin_k[1]), _ptr(lin_v[0]), _ptr(lin_v[1]), _ptr(lin_o[0]), _ptr(lin_o[1]), _ptr(norm1[0]), _ptr(norm1[1]),
          ctypes.c_float(norm1[2]), _ptr(lin_q2[0]), _ptr(lin_q2[1]), _ptr(q1), qp_ptr, wk_ptr, ctypes.c_float(sc), qf_ptr, _stream())
    return q1, out2


def samdec_tok2img_raw_core(qfold, x, pe):
    """The attention core of samdec_tok2img_raw alone: qfold (B,64,256) bf16 folded queries (samdec_tokens_pre(fold=...)), x
    (1|B,N,256) bf16 image tokens, pe (N,256) bf16 or None -> y (B,64,256) f32 = sum_n p_jn x_n (samdec_tokens_post(y=...) applies W_v)."""
    _chk(qfold, torch.bfloat16, "qfold", 3)
    _chk(x, torch.bfloat16, "x", 3)
    B = qfold.shape[0]
    if tuple(qfold.shape[1:]) != (64, 256) or x.shape[0] not in (1, B) or x.shape[2] != 256 or not qfold.is_contiguous():
        raise RuntimeError("samdec_tok2img_raw_core: qfold must be contiguous (B,64,256), x (1|B,N,256) bfloat16")
    if pe is not None:
        _chk(pe, torch.bfloat16, "pe", 2)
    y = torch.empty(B, 64, 256, dtype=torch.float32, device=qfold.device)
    _call("s6d_samdec_tok2img_raw_bf16", _ptr(qfold), _ptr(x), int(x.stride(1)), 1 if x.shape[0] == 1 and B > 1 else 0,
          _ptr(pe) if pe is not None else _vp(0), B, int(x.shape[1]), _ptr(y), _stream())
    return y


def samdec_tokens_post(q1, att, pe, lin_o2, norm2, lin_1, lin_2, norm3, lin_k3, lin_v3, y=None, vfold=None, expand=None):
    """Attention output projection + norm2 + MLP + norm3 + the image->token attention's k / v projections in one launch
    (s6d_samdec_tokens_post_bf16).  q1, pe (B,T,256) f32, att (B,T,128) f32 -> (q3 (B,T,256), kt (B,T,128), vt (B,T,128)) f32.
    y (B,64,256) f32 + vfold = (wvfold, bv): att is formed in the kernel from the attention core's raw result (att = None).
    expand = dict(wq=wqfold, bq=bias, wo=wofold, fold_q=bool): the operands of the image->token attention kernels are made in the
    kernel as well -> (q3, dict(kexp | k256 + cb, vpt)) instead of (q3, kt, vt)."""
    _chk(q1, torch.float32, "q1", 3)
    _chk(pe, torch.float32, "pe", 3)
    B, T, C = q1.shape
    if C != 256 or T > 8 or pe.shape != q1.shape:
        raise ValueError("samdec_tokens_post: q1 / pe must be (B, T <= 8, 256)")
    if y is None:
        _chk(att, torch.float32, "att", 3)
        if tuple(att.shape) != (B, T, 128):
            raise ValueError("samdec_tokens_post: att must be (B, T, 128)")
        att = att.contiguous()
    else:
        _chk(y, torch.float32, "y", 3)
        if tuple(y.shape) != (B, 64, 256) or not y.is_contiguous() or vfold is None:
            raise ValueError("samdec_tokens_post: y must be contiguous (B, 64, 256) with vfold = (wvfold, bv)")
    q1, pe = q1.contiguous(), pe.contiguous()
    dev = q1.device
    q3 = torch.empty(B, T, 256, dtype=torch.float32, device=dev)
    null = _vp(0)
    if expand is None:
        kt = torch.empty(B, T, 128, dtype=torch.float32, device=dev)
        vt = torch.empty(B, T, 128, dtype=torch.float32, device=dev)
        tail = (_ptr(kt), _ptr(vt))
        ex = (null, null, null, null, null, null, null)
        res = (q3, kt, vt)
    else:
        out = {"vpt": torch.empty(B, 256, 64, dtype=torch.bfloat16, device=dev)}
        if expand["fold_q"]:
            out["k256"] = torch.empty(B, 64, 256, dtype=torch.bfloat16, device=dev)
            out["cb"] = torch.empty(B, 64, dtype=torch.float32, device=dev)
            ex = (_ptr(expand["wq"]), _ptr(expand["bq"]), _ptr(expand["wo"]), null, _ptr(out["k256"]), _ptr(out["cb"]), _ptr(out["vpt"]))
        else:
            out["kexp"] = torch.zeros(B, 64, 128, dtype=torch.bfloat16, device=dev)      # block diagonal: the kernel writes the blocks
            ex = (null, null, _ptr(expand["wo"]), _ptr(out["kexp"]), null, null, _ptr(out["vpt"]))
        tail = (null, null)
        res = (q3, out)
    yargs = (null, null, null) if y is None else (_ptr(y), _ptr(vfold[0]), _ptr(vfold[1]))
    _call("s6d_samdec_tokens_post_bf16", _ptr(q1), _ptr(att) if y is None else null, _ptr(pe), B, T, _ptr(lin_o2[0]), _ptr(lin_o2[1]),
          _ptr(norm2[0]), _ptr(norm2[1]), ctypes.c_float(norm2[2]), _ptr(lin_1[0]), _ptr(lin_1[1]), _ptr(lin_2[0]), _ptr(lin_2[1]),
          _ptr(norm3[0]), _ptr(norm3[1]), ctypes.c_float(norm3[2]), _ptr(lin_k3[0]), _ptr(lin_k3[1]), _ptr(lin_v3[0]), _ptr(lin_v3[1]),
          _ptr(q3), *tail, *yargs, *ex, _stream())
    return res


def sam_mask_post(low_res, img_size, input_size, original_size, mask_threshold=0.0, stability_offset=1.0):
    """Fused Sam.postprocess_masks + stability score + threshold + boxes.  low_res (B,C,n,n) f32 -- contiguous, or a channel
    slice ``full[:, c0:c0 + C]`` of a contiguous (B,Ct,n,n) tensor, which is read in place ->
    (masks (B*C,H,W) bool, stability (B*C,) f32, boxes (B*C,4) int64 XYXY, [0,0,0,0] for empty masks)."""
    if not low_res.is_cuda or low_res.dtype != torch.float32 or low_res.dim() != 4:
        raise RuntimeError("low_res must be a 4-d float CUDA tensor")
    B, C, n, n2 = low_res.shape
    if n != n2:
        raise RuntimeError("low_res must be square")
    sb, sc, sy, sx = low_res.stride()
    if not (sx == 1 and sy == n and sc == n * n and sb % (n * n) == 0 and sb >= C * n * n) or B * C == 0:
        low_res = low_res.contiguous()
        sb = C * n * n
    ct = max(sb // (n * n), C, 1)               # planes between consecutive prompts (Ct of the parent tensor for a channel slice)
    Bm, (H, W) = B * C, original_size
    if Bm == 0:                                    # no prompt or no channel: nothing to launch (a clamped channel count would write B rows
        dev = low_res.device                       # into zero-row outputs)
        return (torch.empty(0, H, W, dtype=torch.bool, device=dev), torch.empty(0, dtype=torch.float32, device=dev),
                torch.empty(0, 4, dtype=torch.int64, device=dev))
    masks = torch.empty(Bm, H, W, dtype=torch.bool, device=low_res.device)      # the kernel writes 0 / 1 bytes: no uint8 -> bool pass
    stats = torch.empty(Bm, 6, dtype=torch.int32, device=low_res.device)
    # the slice's data pointer is its first plane: mask (b, c) reads plane b * ct + c from there
    _call("s6d_sam_mask_post_sel_f32", _ptr(low_res), int(B), int(ct), 0, int(max(C, 1)), int(n), int(img_size), int(input_size[0]),
          int(input_size[1]), int(H), int(W), ctypes.c_float(mask_threshold), ctypes.c_float(stability_offset), _ptr(masks),
          _ptr(stats), _stream())
    stability = stats[:, 0] / stats[:, 1]                           # int32 / int32 -> float32, NaN for 0 / 0 like the reference
    empty = (stats[:, 4] < stats[:, 2]) | (stats[:, 5] < stats[:, 3])
    boxes = stats[:, 2:6].long() * (~empty).unsqueeze(-1)
    return masks, stability, boxes


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms: indices of the kept boxes, by decreasing score.  boxes (N,4) f32 XYXY, scores (N,)."""
    _chk(boxes, torch.float32, "boxes", 2)
    N = boxes.shape[0]
    if N == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores.float(), descending=True, stable=True)[1].contiguous()
    fn = _lib.lib().s6d_nms_workspace_bytes
    fn.restype = ctypes.c_long
    ws = torch.empty(max(int(fn(N)), 8), dtype=torch.uint8, device=boxes.device)
    keep = torch.empty(N, dtype=torch.uint8, device=boxes.device)
    _call("s6d_nms_f32", _ptr(boxes), _ptr(order), N, ctypes.c_float(iou_threshold), _ptr(ws), _ptr(keep), _stream())
    return order[keep.bool()]


def samdec_upscale_heads(y0, ln_w, ln_b, eps, w2t, b2, hyper, h, w):
    """Output head of SAM's mask decoder after the first transposed conv (columns ordered (dy,dx,c)).
    y0 (B,h*w,256) bf16 (last-dim slice allowed), w2t (128,64) bf16, hyper (B,M,32) f32 -> masks (B,M,4h,4w) f32."""
    if y0.dtype != torch.bfloat16 or y0.dim() != 3 or y0.shape[-1] != 256 or y0.stride(2) != 1 or not y0.is_cuda:
        raise RuntimeError("y0 must be a (B,h*w,256) bfloat16 CUDA tensor with unit channel stride")
    B, N = y0.shape[0], y0.shape[1]
    y_ld = y0.stride(1)
    if B > 1 and y0.stride(0) != N * y_ld:
        raise RuntimeError("y0 batch stride must be N * row stride")
    _chk(w2t, torch.bfloat16, "w2t", 2)
    _chk(hyper, torch.float32, "hyper", 3)
    for t, nm in ((ln_w, "ln_w"), (ln_b, "ln_b"), (b2, "b2")):
        _chk(t, torch.float32, nm, 1)
    M = hyper.shape[1]
    if N != h * w or tuple(w2t.shape) != (128, 64) or hyper.shape[2] != 32 or hyper.shape[0] != B:
        raise RuntimeError("samdec_upscale_heads: shape mismatch")
    masks = torch.empty(B, M, 4 * h, 4 * w, dtype=torch.float32, device=y0.device)
    _call("s6d_samdec_upscale_heads_bf16", _ptr(y0), _ptr(ln_w), _ptr(ln_b), ctypes.c_float(eps), _ptr(w2t), _ptr(b2),
          _ptr(hyper), B, M, int(h), int(w), int(y_ld), _ptr(masks), _stream())
    return masks


def segment_seq_sum(x, start, count):
    """Row-order float32 sums of segments of x (numpy's add.reduce over axis 0; the centroid numerator of the PEM
    pre-processing).  x (N,C) f32 with 2 <= C <= 4, start / count (P,) int64 -> (P,C) f32."""
    _chk(x, torch.float32, "x", 2)
    _chk(start, torch.int64, "start", 1)
    _chk(count, torch.int64, "count", 1)
    P, C = start.shape[0], x.shape[1]
    if count.shape[0] != P or not 2 <= C <= 4:
        raise RuntimeError("segment_seq_sum: start/count must have one entry per segment and C must be 2..4")
    out = torch.zeros(P, C, dtype=torch.float32, device=x.device)
    _call("s6d_segment_seq_sum_f32", _ptr(x), _ptr(start), _ptr(count), P, C, _ptr(out), _stream())
    return out


def pem_sample_indices(keys, count, n_sample):
    """Sampler of the PEM pre-processing (one workgroup per detection: histogram threshold + in-LDS sort).  keys (P,L) f32
    uniforms in [0,1), count (P,) int64 (<= L) -> (idx (P,n_sample) int64, overflow (P,) int32)."""
    _chk(keys, torch.float32, "keys", 2)
    _chk(count, torch.int64, "count", 1)
    P, L = keys.shape
    if count.shape[0] != P:
        raise RuntimeError("pem_sample_indices: one count per row of keys")
    idx = torch.zeros(P, n_sample, dtype=torch.int64, device=keys.device)
    overflow = torch.zeros(P, dtype=torch.int32, device=keys.device)
    _call("s6d_pem_sample_indices_f32", _ptr(keys), ctypes.c_long(L), _ptr(count), P, int(n_sample), _ptr(idx), _ptr(overflow),
          _stream())
    return idx, overflow


def pem_mask_boxes(mask8, depth, min_points):
    """mask8 (P,H,W) uint8 (non-zero = set), depth (H,W) f32 -> (m8 (P,H,W) uint8 = mask AND depth > 0, cnt (P,) int64,
    ok8 (P,) uint8 = cnt > min_points, box (P,4) int64 [y1,y2,x1,x2] by the reference's get_bbox)."""
    _chk(mask8, torch.uint8, "mask8", 3)
    _chk(depth, torch.float32, "depth", 2)
    P, H, W = mask8.shape
    if tuple(depth.shape) != (H, W):
        raise RuntimeError("pem_mask_boxes: depth must be (H,W)")
    dev = mask8.device
    m8 = torch.empty(P, H, W, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(P, dtype=torch.int64, device=dev)
    ok8 = torch.zeros(P, dtype=torch.uint8, device=dev)
    box = torch.zeros(P, 4, dtype=torch.int64, device=dev)
    _call("s6d_pem_mask_boxes_u8", _ptr(mask8), _ptr(depth), P, H, W, ctypes.c_long(int(min_points)), _ptr(m8), _ptr(cnt), _ptr(ok8),
          _ptr(box), _stream())
    return m8, cnt, ok8, box


def pem_crops(image_u8, m8, kept, box, S, use_mask, mean, std):
    """image (H,W,3) uint8, m8 (P,H,W) uint8, kept (M,) int64, box (P,4) int64 -> (M,3,S,S) f32 normalised masked crops."""
    _chk(image_u8, torch.uint8, "image", 3)
    _chk(m8, torch.uint8, "m8", 3)
    _chk(kept, torch.int64, "kept", 1)
    _chk(box, torch.int64, "box", 2)
    P, H, W = m8.shape
    if tuple(image_u8.shape) != (H, W, 3) or tuple(box.shape) != (P, 4):
        raise RuntimeError("pem_crops: shape mismatch")
    M = kept.shape[0]
    out = torch.empty(M, 3, S, S, dtype=torch.float32, device=m8.device)
    _call("s6d_pem_crops_f32", _ptr(image_u8), _ptr(m8), _ptr(kept), _ptr(box), M, H, W, int(S), int(bool(use_mask)),
          (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std), _ptr(out), _stream())
    return out


def pem_compact_cloud(m8, depth, box, ok8, fx, fy, cx, cy, cap):
    """m8 (P,H,W) uint8, depth (H,W) f32, box (P,4) int64, ok8 (P,) uint8 -> (choose (P,cap) int32, cloud (P,cap,3) f32,
    n (P,) int64): masked crop pixels in row-major crop order and their back-projections, one fixed-capacity slot each."""
    _chk(m8, torch.uint8, "m8", 3)
    _chk(depth, torch.float32, "depth", 2)
    _chk(box, torch.int64, "box", 2)
    _chk(ok8, torch.uint8, "ok8", 1)
    P, H, W = m8.shape
    if tuple(depth.shape) != (H, W) or tuple(box.shape) != (P, 4) or ok8.shape[0] != P:
        raise RuntimeError("pem_compact_cloud: shape mismatch")
    choose = torch.empty(P, cap, dtype=torch.int32, device=m8.device)
    cloud = torch.empty(P, cap, 3, dtype=torch.float32, device=m8.device)
    n = torch.zeros(P, dtype=torch.int64, device=m8.device)
    _call("s6d_pem_compact_cloud_f32", _ptr(m8), _ptr(depth), _ptr(box), _ptr(ok8), P, H, W, ctypes.c_float(fx), ctypes.c_float(fy),
          ctypes.c_float(cx), ctypes.c_float(cy), ctypes.c_long(cap), _ptr(choose), _ptr(cloud), _ptr(n), _stream())
    return choose, cloud, n


def pem_radius_filter(center, limit, choose, cloud, n):
    """In place on (choose, cloud, n) of pem_compact_cloud: keep, in order, the points within limit[p] of center[p]."""
    _chk(center, torch.float32, "center", 2)
    _chk(limit, torch.float64, "limit", 1)
    _chk(choose, torch.int32, "choose", 2)
    _chk(cloud, torch.float32, "cloud", 3)
    _chk(n, torch.int64, "n", 1)
    P, cap = choose.shape
    if tuple(center.shape) != (P, 3) or limit.shape[0] != P or tuple(cloud.shape) != (P, cap, 3) or n.shape[0] != P:
        raise RuntimeError("pem_radius_filter: shape mismatch")
    _call("s6d_pem_radius_filter_f32", _ptr(center), _ptr(limit), P, ctypes.c_long(cap), _ptr(choose), _ptr(cloud), _ptr(n), _stream())
    return n


def upsample_gather(up, choose, H, W, C):
    """up (B,196,16*C) f32, choose (B,n) int64 -> (B,n,C): bilinear x4 of the pixel-shuffled map at chosen pixels."""
    _chk(up, torch.float32, "up", 3)
    _chk(choose, torch.int64, "choose", 2)
    B, T, PC = up.shape
    G = int(round(T ** 0.5))
    P = int(round((PC // C) ** 0.5))
    n = choose.shape[1]
    out = torch.empty(B, n, C, dtype=torch.float32, device=up.device)
    _call("s6d_upsample_gather_f32", _ptr(up), _ptr(choose), B, n, G, P, int(C), int(H), int(W), _ptr(out), _stream())
    return out


# ------------------------------------------------------------------ SAM image encoder
_REL_PAD = {}


def _padded_rel(rel_h, rel_w, H, window, hd):
    """The zero-padded copies of a (rel_h, rel_w) table pair in the layout the attention kernels read (s6d_win_attention_pad_rel_bf16),
    cached on the identity and version of the two tensors (the modules hand in their own cached bf16 tables)."""
    key = (id(rel_h), id(rel_w), H, window, hd)
    c = _REL_PAD.get(key)
    if c is None or c[0] is not rel_h or c[1] is not rel_w or c[2] != (rel_h._version, rel_w._version):
        fn = _lib.lib().s6d_win_attention_scratch_bytes
        fn.restype = ctypes.c_long
        pad = torch.empty(int(fn(H, window, hd)), dtype=torch.uint8, device=rel_h.device)
        _call("s6d_win_attention_pad_rel_bf16", _ptr(rel_h), _ptr(rel_w), H, window, hd, _ptr(pad), _stream())
        if len(_REL_PAD) > 256:
            _REL_PAD.clear()
        c = (rel_h, rel_w, (rel_h._version, rel_w._version), pad)
        _REL_PAD[key] = c
    return c[3]


def window_attention(qkv, qkv_bias, rel_h, rel_w, num_heads, window, scale, head_major_shape=None):
    """qkv (B,H,W,3C) bf16, qkv_bias (3C) bf16, rel_h/rel_w (2S-1,hd) bf16 or None -> (B,H,W,C) bf16.
    head_major_shape=(B,H,W): qkv is the head-major tensor (3*num_heads, B*H*W, hd) that gemm_bf16(..., col_block=hd) writes."""
    _chk(qkv_bias, torch.bfloat16, "qkv_bias", 1)
    if head_major_shape is None:
        _chk(qkv, torch.bfloat16, "qkv", 4)
        B, H, W, C3 = qkv.shape
        C = C3 // 3
        hd = C // num_heads
    else:
        _chk(qkv, torch.bfloat16, "qkv", 3)
        B, H, W = head_major_shape
        hd = qkv.shape[2]
        C = num_heads * hd
        if tuple(qkv.shape) != (3 * num_heads, B * H * W, hd):
            raise RuntimeError(f"head-major qkv must be (3*heads, B*H*W, hd); got {tuple(qkv.shape)} for B,H,W={head_major_shape}")
    if rel_h is not None:
        _chk(rel_h, torch.bfloat16, "rel_h", 2)
        _chk(rel_w, torch.bfloat16, "rel_w", 2)
        S = window if window > 0 else H
        if rel_h.shape != (2 * S - 1, hd) or rel_w.shape != (2 * S - 1, hd):
            raise RuntimeError("rel_pos tables must be (2S-1, head_dim)")
    out = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=qkv.device)
    if rel_h is not None:
        # the padded copies of the two tables the kernels read are a function of the tables only: made once per (table pair, grid,
        # window) and kept (round 6: the padding ran as a 5-us launch in front of each of the 64 attention launches of a step)
        _call("s6d_win_attention_prepadded_bf16", _ptr(qkv), 0 if head_major_shape is None else 1, _ptr(qkv_bias),
              _ptr(_padded_rel(rel_h, rel_w, H, int(window), int(hd))), B, H, W, int(num_heads), int(hd), int(window), ctypes.c_float(scale),
              _ptr(out), _stream())
        return out
    _call("s6d_win_attention_layout_bf16", _ptr(qkv), 0 if head_major_shape is None else 1, _ptr(qkv_bias), _vp(0), _vp(0), B, H, W,
          int(num_heads), int(hd), int(window), ctypes.c_float(scale), _vp(0), _ptr(out), _stream())
    return out


def _half(t, name, ndim=None):
    """bf16 or IEEE-half tensor -> symbol suffix of the kernel family compiled for that element type."""
    if t.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError(f"{name} must be a bfloat16 or float16 tensor")
    _chk(t, t.dtype, name, ndim)
    return "bf16" if t.dtype == torch.bfloat16 else "f16"


def seq_attention(qkv, num_heads, scale, seq_len=None):
    """softmax(scale q k^T) v per head, no positional bias.  qkv (B,N,3C) bf16 or f16 token-major (the raw Linear output), or --
    with ``seq_len`` = N -- HEAD-major (3 * num_heads, B * N, hd), what ``gemm_bf16(..., col_block=hd)`` / ``gemm_bf16_lnfold(...,
    col_block=hd)`` return: the K / V rows of a (sequence, head) are then one contiguous run instead of N pieces of hd elements
    strided by 3C (the fetch of those pieces alone costs 136 us per 150 x 16 x 257 launch).  -> (B,N,C) in qkv's dtype."""
    sfx = _half(qkv, "qkv", 3)
    if seq_len is None:
        B, N, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty(B, N, C, dtype=qkv.dtype, device=qkv.device)
        _call("s6d_seq_attention_" + sfx, _ptr(qkv), B, N, int(num_heads), int(C // num_heads), ctypes.c_float(scale),
              _ptr(out), _stream())
        return out
    H3, M, hd = qkv.shape
    N = int(seq_len)
    if H3 != 3 * num_heads or M % N:
        raise ValueError(f"head-major qkv must be (3 * num_heads, B * N, hd); got {tuple(qkv.shape)} for {num_heads} heads, N = {N}")
    B = M // N
    out = torch.empty(B, N, num_heads * hd, dtype=qkv.dtype, device=qkv.device)
    _call("s6d_seq_attention_strided_" + sfx, _ptr(qkv), ctypes.c_long(hd), ctypes.c_long(num_heads * M * hd), ctypes.c_long(M * hd),
          B, N, int(num_heads), int(hd), ctypes.c_float(scale), _ptr(out), _stream())
    return out


def gemm_bf16(a, w, bias=None, gelu=False, out=None, max_blocks=0, col_block=0, residual=None, stats_partial=None):
    """a (..., K) bf16 (rows may be strided), w (N, K) bf16 = nn.Linear.weight, bias (N) f32 or None ->
    act(a @ w.T + bias) (..., N) bf16 with act = exact GELU or identity; N % 128 == 0, K % 64 == 0.
    col_block > 0: the output comes back as (N / col_block, M, col_block) -- column blocks as separate matrices (N % 256 == 0).
    residual (..., N) bf16: returns bf16(a @ w.T + bias + residual), summed in fp32 (N % 256 == 0, no GELU); out may be the residual
    itself.  stats_partial (N / 32, 2, M) f32 (with residual): receives the partial LayerNorm statistics of the result rows
    (ln_stats_finalize turns them into (mean, rstd))."""
    if not a.is_cuda or not w.is_cuda:
        raise RuntimeError("a and w must be CUDA tensors")
    if a.dtype not in (torch.bfloat16, torch.float16) or w.dtype != a.dtype:
        raise RuntimeError("a and w must both be bfloat16 (or both float16) tensors")
    f16 = a.dtype == torch.float16          # IEEE half: the same kernel on v_mfma_f32_32x32x16_f16 (plain / GELU epilogue, N % 256 == 0)
    if f16 and (col_block or residual is not None):
        raise RuntimeError("the float16 GEMM has the plain and the GELU epilogue only")
    K = a.shape[-1]
    N = w.shape[0]
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 8 or a2.data_ptr() % 16:
        a2 = a2.contiguous()
    if w.dim() != 2 or w.shape[1] != K or w.stride(1) != 1 or w.stride(0) % 8:
        raise RuntimeError("w must be (N, K) with contiguous rows")
    if bias is not None:
        _chk(bias, torch.float32, "bias", 1)
    M = a2.shape[0]
    if col_block:
        if out is not None:
            raise RuntimeError("col_block output is allocated by the call")
        out = torch.empty(N // col_block, M, col_block, dtype=torch.bfloat16, device=a.device)   # (bf16 only: checked above)
        _call("s6d_gemm_bf16_cblk", _ptr(a2), ctypes.c_long(a2.stride(0)), _ptr(w), ctypes.c_long(w.stride(0)),
              _ptr(bias) if bias is not None else _vp(0), _ptr(out), ctypes.c_long(N), M, N, K, 1 if gelu else 0, int(col_block),
              int(max_blocks), _stream())
        return out
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    else:
        _chk(out, a.dtype, "out", 2)
        if tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.stride(0) % 8 or out.data_ptr() % 16:
            raise ValueError(f"out must be ({M}, {N}) bf16 with contiguous, 16-byte aligned rows (stride % 8 == 0); got "
                             f"{tuple(out.shape)} strides {tuple(out.stride())}")
    # the kernel's staging addresses are 32-bit byte offsets from A: rows go through in slabs below 2 GiB (ViT-H lin2 reaches
    # the limit at 52 frames per call)
    rows = gemm_one_launch_rows(a2.stride(0))
    if residual is not None:
        if gelu or col_block:
            raise RuntimeError("the residual form takes neither GELU nor column blocks")
        _chk(residual, torch.bfloat16, "residual")
        r2 = residual.reshape(-1, N)
        if r2.shape[0] != M:
            raise ValueError(f"residual has {r2.shape[0]} rows, the product {M}")
        if r2.stride(1) != 1 or r2.stride(0) % 8 or r2.data_ptr() % 16:
            raise ValueError("residual rows must be contiguous and 16-byte aligned")
        if stats_partial is not None:
            _chk(stats_partial, torch.float32, "stats_partial")
            if tuple(stats_partial.shape) != (N // 32, 2, M) or not stats_partial.is_contiguous():
                raise ValueError(f"stats_partial must be a contiguous ({N // 32}, 2, {M}) f32 tensor; got {tuple(stats_partial.shape)}")
            if M > rows:
                raise RuntimeError("row statistics are written for one launch: the row count exceeds one 2-GiB slab")
        for r0 in range(0, M, rows):
            r1 = min(M, r0 + rows)
            _call("s6d_gemm_bf16_res", _ptr(a2[r0:r1]), ctypes.c_long(a2.stride(0)), _ptr(w), ctypes.c_long(w.stride(0)),
                  _ptr(bias) if bias is not None else _vp(0), _ptr(r2[r0:r1]), ctypes.c_long(r2.stride(0)),
                  _ptr(stats_partial) if stats_partial is not None else _vp(0), _ptr(out[r0:r1]),
                  ctypes.c_long(out.stride(0)), r1 - r0, N, K, int(max_blocks), _stream())
        return out.reshape(*a.shape[:-1], N)
    if stats_partial is not None:
        raise RuntimeError("stats_partial comes with the residual form")
    for r0 in range(0, M, rows):
        r1 = min(M, r0 + rows)
        _call("s6d_gemm_f16" if f16 else "s6d_gemm_bf16", _ptr(a2[r0:r1]), ctypes.c_long(a2.stride(0)), _ptr(w), ctypes.c_long(w.stride(0)),
              _ptr(bias) if bias is not None else _vp(0), _ptr(out[r0:r1]), ctypes.c_long(out.stride(0)), r1 - r0, N, K,
              1 if gelu else 0, int(max_blocks), _stream())
    return out.reshape(*a.shape[:-1], N)


def set_geo_embed_form(form):
    """Which kernel serves the pre-split geometric embedding (include/sam6d_hip.h: s6d_set_geo_embed_form): 1 = the two-phase kernel
    of rounds 3-5 (default), 2 = registers-built sinusoid fragments + LDS-DMA weight slices (round 6).  Same bits."""
    _call("s6d_set_geo_embed_form", int(form))


def set_gemm_small_tile(enable):
    """256 x 128 tiles for under-filled plain / GELU GEMM launches (include/sam6d_hip.h: s6d_set_gemm_small_tile); same bits either way."""
    _call("s6d_set_gemm_small_tile", int(enable))                  # False / True / 2 (= also the residual epilogue)


def set_gemm_wave_tile(columns):
    """Which form of the bf16 / f16 GEMM kernel serves the shapes both cover (include/sam6d_hip.h: s6d_set_gemm_wave_tile): 0 = the
    library's choice per shape, 64 = the eight-wave form, 128 = the four-wave form wherever it applies.  Process-wide; for A/B
    measurements and the parity tests (the two forms give the same bits)."""
    _call("s6d_set_gemm_wave_tile", int(columns))


def gemm_one_launch_rows(row_stride, elem_bytes=2):
    """Rows ONE launch of the bf16 / f16 GEMM kernels takes for an A operand of ``row_stride`` elements per row: their staging
    addresses are 32-bit byte offsets from A, so a launch stays below 2 GiB, in whole 256-row tiles.  The plain forms walk larger
    inputs in slabs of this many rows; the forms that carry row statistics (residual + stats_partial, lnfold) are one launch, so
    their callers' eligibility checks use this very number (one helper instead of three bounds)."""
    return max(256, ((2 ** 31 - 1) // (elem_bytes * int(row_stride))) // 256 * 256)


def ln_stats_finalize(stats_partial, group_size=32, eps=1e-6):
    """(groups, 2, M) f32 partial statistics (gemm_bf16(..., stats_partial=)) -> (M, 2) f32 rows of (mean, sigma = sqrt(var + eps))."""
    _chk(stats_partial, torch.float32, "stats_partial", 3)
    if not stats_partial.is_contiguous() or stats_partial.shape[1] != 2:
        raise ValueError("stats_partial must be a contiguous (groups, 2, M) tensor")
    G, _, M = stats_partial.shape
    out = torch.empty(M, 2, dtype=torch.float32, device=stats_partial.device)
    _call("s6d_ln_stats_finalize", _ptr(stats_partial), int(G), int(group_size), ctypes.c_long(M), ctypes.c_float(eps), _ptr(out), _stream())
    return out


def row_stats(x, eps=1e-6):
    """x (..., C) bf16 -> (rows, 2) f32 of (mean, sigma = sqrt(var + eps)) over the last dim: the LayerNorm statistics alone."""
    _chk(x, torch.bfloat16, "x")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    out = torch.empty(x2.shape[0], 2, dtype=torch.float32, device=x.device)
    _call("s6d_row_stats_bf16", _ptr(x2), ctypes.c_long(x2.stride(0)), ctypes.c_long(x2.shape[0]), int(C), ctypes.c_float(eps), _ptr(out),
          _stream())
    return out


def gemm_bf16_lnfold(a, stats, w_folded, col_sums, bias, gelu=False, col_block=0, max_blocks=0):
    """act(LayerNorm(a) @ W.T + b) without the normalised activations: a (..., K) bf16 raw rows, stats (rows, 2) f32 their
    (mean, sigma), w_folded (N, K) bf16 = gamma * W, col_sums (N) f32 = w_folded.float().sum(1), bias (N) f32 = b + W @ beta
    (sam6d_amd/utils/linear.py::lnfold_weights).  N % 256 == 0.  col_block as gemm_bf16."""
    _chk(a, torch.bfloat16, "a")
    _chk(w_folded, torch.bfloat16, "w_folded", 2)
    _chk(stats, torch.float32, "stats", 2)
    _chk(col_sums, torch.float32, "col_sums", 1)
    _chk(bias, torch.float32, "bias", 1)
    K, N = a.shape[-1], w_folded.shape[0]
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 8 or a2.data_ptr() % 16:
        a2 = a2.contiguous()
    M = a2.shape[0]
    if w_folded.shape[1] != K or w_folded.stride(1) != 1 or w_folded.stride(0) % 8:
        raise RuntimeError("w_folded must be (N, K) with contiguous rows")
    if tuple(stats.shape) != (M, 2) or not stats.is_contiguous() or col_sums.numel() != N or bias.numel() != N:
        raise ValueError(f"stats must be ({M}, 2), col_sums and bias ({N},)")
    if M > gemm_one_launch_rows(a2.stride(0)):
        raise RuntimeError("one launch: the row count exceeds one 2-GiB slab")
    if col_block:
        out = torch.empty(N // col_block, M, col_block, dtype=torch.bfloat16, device=a.device)
        ldc = N
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
        ldc = N
    _call("s6d_gemm_bf16_lnfold", _ptr(a2), ctypes.c_long(a2.stride(0)), _ptr(stats), _ptr(w_folded), ctypes.c_long(w_folded.stride(0)),
          _ptr(col_sums), _ptr(bias), _ptr(out), ctypes.c_long(ldc), M, N, K, 1 if gelu else 0, int(col_block), int(max_blocks), _stream())
    return out if col_block else out.reshape(*a.shape[:-1], N)


def gemm_fp8(a8, a_scale, w8, w_scale, bias=None, gelu=False, max_blocks=0):
    """a8 (..., K) uint8 (e4m3fn bytes) with a_scale (rows,) uint8 (E8M0), w8 (N, K) uint8 with w_scale (N,) uint8, bias (N) f32
    or None -> act(A W^T + bias) (..., N) bf16 on the fp8 matrix cores (sam6d_amd/utils/fp8.py: the operand format).
    N % 256 == 0, K % 128 == 0."""
    for t, nm in ((a8, "a8"), (a_scale, "a_scale"), (w8, "w8"), (w_scale, "w_scale")):
        _chk(t, torch.uint8, nm)
    K, N = a8.shape[-1], w8.shape[0]
    a2 = a8.reshape(-1, K)
    M = a2.shape[0]
    if w8.shape != (N, K) or a_scale.numel() != M or w_scale.numel() != N:
        raise ValueError(f"shapes: a8 {tuple(a8.shape)}, a_scale {tuple(a_scale.shape)}, w8 {tuple(w8.shape)}, w_scale {tuple(w_scale.shape)}")
    if bias is not None:
        _chk(bias, torch.float32, "bias", 1)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a8.device)
    rows = max(256, ((2 ** 31 - 1) // K) // 256 * 256)
    for r0 in range(0, M, rows):
        r1 = min(M, r0 + rows)
        _call("s6d_gemm_fp8", _ptr(a2[r0:r1]), ctypes.c_long(K), _ptr(a_scale.reshape(-1)[r0:r1]), _ptr(w8), ctypes.c_long(K),
              _ptr(w_scale), _ptr(bias) if bias is not None else _vp(0), _ptr(out[r0:r1]), ctypes.c_long(N), r1 - r0, N, K,
              1 if gelu else 0, int(max_blocks), _stream())
    return out.reshape(*a8.shape[:-1], N)


def gemm_fp8_gelu_mx(a8, a_scale, w8, w_scale, bias=None, max_blocks=0):
    """lin1 of the fp8 block with an MX output: a8 (..., K) uint8 e4m3 + a_scale (rows,) uint8, w8 (N, K) + w_scale (N,) ->
    (q (rows, N) uint8 = e4m3(GELU(A W^T + bias)), s (rows, N / 32) uint8 = one E8M0 scale per row and 32 columns): the MX A operand
    of gemm_fp8_mxa.  N % 256 == 0, K % 128 == 0; one launch (rows * K < 2^31)."""
    for t, nm in ((a8, "a8"), (a_scale, "a_scale"), (w8, "w8"), (w_scale, "w_scale")):
        _chk(t, torch.uint8, nm)
    K, N = a8.shape[-1], w8.shape[0]
    a2 = a8.reshape(-1, K)
    M = a2.shape[0]
    if w8.shape != (N, K) or a_scale.numel() != M or w_scale.numel() != N:
        raise ValueError(f"shapes: a8 {tuple(a8.shape)}, a_scale {tuple(a_scale.shape)}, w8 {tuple(w8.shape)}, w_scale {tuple(w_scale.shape)}")
    if bias is not None:
        _chk(bias, torch.float32, "bias", 1)
    if M * K >= 2 ** 31:
        raise RuntimeError("one launch: the row count exceeds one 2-GiB slab")
    q = torch.empty(M, N, dtype=torch.uint8, device=a8.device)
    Mp = (M + 255) // 256 * 256                 # the consumer fetches the scale dwords of whole 256-row tiles: the rows exist
    s = torch.zeros(Mp, N // 32, dtype=torch.uint8, device=a8.device)[:M]
    _call("s6d_gemm_fp8_gelu_mx", _ptr(a2), ctypes.c_long(K), _ptr(a_scale.reshape(-1)), _ptr(w8), ctypes.c_long(K), _ptr(w_scale),
          _ptr(bias) if bias is not None else _vp(0), _ptr(q), ctypes.c_long(N), _ptr(s), M, N, K, int(max_blocks), _stream())
    return q, s


def gemm_fp8_mxa(a8, a_mx, w8, w_scale, bias=None, gelu=False, max_blocks=0):
    """a8 (rows, K) uint8 e4m3 with a_mx (rows, K / 32) uint8 MX block scales (gemm_fp8_gelu_mx's outputs), w8 (N, K) uint8 with
    w_scale (N,) -> act(A W^T + bias) (rows, N) bf16.  rows % 256 == 0, N % 256 == 0, K % 128 == 0."""
    for t, nm in ((a8, "a8"), (a_mx, "a_mx"), (w8, "w8"), (w_scale, "w_scale")):
        _chk(t, torch.uint8, nm)
    K, N = a8.shape[-1], w8.shape[0]
    a2 = a8.reshape(-1, K)
    M = a2.shape[0]
    if w8.shape != (N, K) or tuple(a_mx.shape) != (M, K // 32) or w_scale.numel() != N:
        raise ValueError(f"shapes: a8 {tuple(a8.shape)}, a_mx {tuple(a_mx.shape)}, w8 {tuple(w8.shape)}, w_scale {tuple(w_scale.shape)}")
    if bias is not None:
        _chk(bias, torch.float32, "bias", 1)
    if M * K >= 2 ** 31:
        raise RuntimeError("one launch: the row count exceeds one 2-GiB slab")
    Mp = (M + 255) // 256 * 256
    if a_mx.untyped_storage().nbytes() - a_mx.storage_offset() < Mp * (K // 32):
        pad = torch.zeros(Mp, K // 32, dtype=torch.uint8, device=a8.device)
        pad[:M] = a_mx
        a_mx = pad[:M]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a8.device)
    _call("s6d_gemm_fp8_mxa", _ptr(a2), ctypes.c_long(K), _ptr(a_mx), _ptr(w8), ctypes.c_long(K), _ptr(w_scale),
          _ptr(bias) if bias is not None else _vp(0), _ptr(out), ctypes.c_long(N), M, N, K, 1 if gelu else 0, int(max_blocks), _stream())
    return out


def layernorm_fp8(x, gamma, beta, eps, delta=None):
    """x (..., C) bf16 -> (LN(x) as e4m3fn bytes (..., C) uint8, one E8M0 scale byte per row (rows,) uint8).
    delta (same shape, bf16): -> (x + delta, bytes, scales) with the residual add folded in, as add_layernorm."""
    _chk(x, torch.bfloat16, "x")
    _chk(gamma, torch.float32, "gamma", 1)
    _chk(beta, torch.float32, "beta", 1)
    C = x.shape[-1]
    if gamma.numel() != C or beta.numel() != C:
        raise ValueError(f"gamma / beta must have {C} elements")
    if delta is not None and delta.shape != x.shape:
        raise ValueError("delta must have x's shape")
    rows = x.numel() // C
    y8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    ys = torch.empty(rows, dtype=torch.uint8, device=x.device)
    if delta is None:
        _call("s6d_layernorm_fp8", _ptr(x), _ptr(gamma), _ptr(beta), ctypes.c_float(eps), ctypes.c_long(rows), int(C), _ptr(y8),
              _ptr(ys), _stream())
        return y8, ys
    _chk(delta, torch.bfloat16, "delta")
    xo = torch.empty_like(x)
    _call("s6d_add_layernorm_fp8", _ptr(x), _ptr(delta), _ptr(gamma), _ptr(beta), ctypes.c_float(eps), ctypes.c_long(rows), int(C),
          _ptr(xo), _ptr(y8), _ptr(ys), _stream())
    return xo, y8, ys


def split_weight(w):
    """W (N,K) f32 -> (hi, lo) bf16 parts for linear_f32 (W = hi + lo up to 2^-17 relative)."""
    _chk(w, torch.float32, "w")
    hi = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
    lo = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
    _call("s6d_linear_split_weight_f32", _ptr(w), ctypes.c_long(w.numel()), _ptr(hi), _ptr(lo), _stream())
    return hi, lo


def linear_f32(x, w_hi, w_lo, bias=None, relu=False, residual=None, ln=None):
    """y = LN(residual + act(x W^T + bias)) in one kernel: x (..., K) f32, (w_hi, w_lo) = split_weight(W (N,K)), bias (N) or None,
    residual (..., N) or None, ln = (gamma, beta, eps) or None (N == 256).  K % 32 == 0, N % 256 == 0."""
    _chk(x, torch.float32, "x")
    _chk(w_hi, torch.bfloat16, "w_hi", 2)
    _chk(w_lo, torch.bfloat16, "w_lo", 2)
    if w_lo.shape != w_hi.shape:
        raise ValueError(f"w_hi {tuple(w_hi.shape)} and w_lo {tuple(w_lo.shape)} differ in shape")
    N, K = w_hi.shape
    if x.shape[-1] != K:
        raise ValueError(f"x has {x.shape[-1]} columns, W {K}")
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    r2 = None
    if residual is not None:
        _chk(residual, torch.float32, "residual")
        r2 = residual.reshape(-1, N)
        if r2.shape[0] != M:
            raise ValueError("residual rows != x rows")
    g = bt = None
    eps = 0.0
    if ln is not None:
        g, bt, eps = ln
        _chk(g, torch.float32, "gamma", 1)
        _chk(bt, torch.float32, "beta", 1)
        if g.numel() != N or bt.numel() != N:
            raise ValueError(f"gamma / beta must have {N} elements")
    if bias is not None:
        _chk(bias, torch.float32, "bias", 1)
        if bias.numel() != N:
            raise ValueError(f"bias must have {N} elements")
    _call("s6d_linear_f32", _ptr(x2), ctypes.c_long(x2.stride(0)), M, K, _ptr(w_hi), _ptr(w_lo), _ptr(bias) if bias is not None else _vp(0),
          N, 1 if relu else 0, _ptr(r2) if r2 is not None else _vp(0), ctypes.c_long(r2.stride(0) if r2 is not None else 0),
          _ptr(g) if g is not None else _vp(0), _ptr(bt) if bt is not None else _vp(0), ctypes.c_float(eps), _ptr(y), ctypes.c_long(N),
          _stream())
    return y.reshape(*x.shape[:-1], N)


def fragment_weight(w):
    """w (N, K) bf16 (a part of split_weight's output) -> the same elements in matrix-instruction fragment order (s6d_linear_fragment_weight),
    the weight format of attn_output_chain."""
    _chk(w, torch.bfloat16, "w", 2)
    w = w.contiguous()
    out = torch.empty_like(w)
    _call("s6d_linear_fragment_weight", _ptr(w), int(w.shape[0]), int(w.shape[1]), _ptr(out), _stream())
    return out


def attn_output_chain(a, x, w1, ln1, we, ws, ln2):
    """LN2(h + relu(h We^T + be) Ws^T + bs) with h = LN1(x + a W1^T + b1) in one kernel (s6d_attn_output_chain_f32): a, x (..., 256)
    f32; w1 / we / ws = (w_hi, w_lo, bias) of the (256,256) / (512,256) / (256,512) Linear layers (split_weight, each part through
    fragment_weight); ln1 / ln2 =
    (gamma, beta, eps).  Equals linear_f32 x 3 bit for bit."""
    _chk(a, torch.float32, "a")
    _chk(x, torch.float32, "x")
    if a.shape != x.shape or a.shape[-1] != 256:
        raise ValueError(f"a {tuple(a.shape)} and x {tuple(x.shape)} must be (..., 256)")
    for (hi, lo, b), shp in ((w1, (256, 256)), (we, (512, 256)), (ws, (256, 512))):
        _chk(hi, torch.bfloat16, "w_hi", 2)
        _chk(lo, torch.bfloat16, "w_lo", 2)
        _chk(b, torch.float32, "bias", 1)
        if tuple(hi.shape) != shp or tuple(lo.shape) != shp or b.numel() != shp[0] or not (hi.is_contiguous() and lo.is_contiguous()):
            raise ValueError(f"weight parts must be contiguous {shp}")
    for g, bt, _ in (ln1, ln2):
        _chk(g, torch.float32, "gamma", 1)
        _chk(bt, torch.float32, "beta", 1)
    a2, x2 = a.reshape(-1, 256), x.reshape(-1, 256)
    if a2.stride(1) != 1 or a2.stride(0) % 4 or a2.data_ptr() % 16:
        a2 = a2.contiguous()
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    M = a2.shape[0]
    y = torch.empty(M, 256, dtype=torch.float32, device=a.device)
    _call("s6d_attn_output_chain_f32", _ptr(a2), ctypes.c_long(a2.stride(0)), _ptr(x2), ctypes.c_long(x2.stride(0)), M,
          _ptr(w1[0]), _ptr(w1[1]), _ptr(w1[2]), _ptr(ln1[0]), _ptr(ln1[1]), ctypes.c_float(ln1[2]),
          _ptr(we[0]), _ptr(we[1]), _ptr(we[2]), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ln2[0]), _ptr(ln2[1]), ctypes.c_float(ln2[2]),
          _ptr(y), ctypes.c_long(256), _stream())
    return y.reshape(a.shape)


def layernorm_f32out(x, gamma, beta, eps):
    """x (..., C) bf16 -> LN(x) (..., C) float32 (fp32 statistics, no rounding of the result)."""
    _chk(x, torch.bfloat16, "x")
    _chk(gamma, torch.float32, "gamma", 1)
    _chk(beta, torch.float32, "beta", 1)
    C = x.shape[-1]
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _call("s6d_layernorm_bf16_f32", _ptr(x), _ptr(gamma), _ptr(beta), ctypes.c_float(eps), ctypes.c_long(x.numel() // C), int(C),
          _ptr(y), _stream())
    return y


def add_layernorm(x, delta, gamma, beta, eps):
    """x (...,C) bf16 or f16, delta same shape / dtype or None, gamma/beta (C) f32 -> (x + delta, LN(x + delta)) in x's dtype."""
    sfx = _half(x, "x")
    _chk(gamma, torch.float32, "gamma", 1)
    _chk(beta, torch.float32, "beta", 1)
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    if delta is not None:
        _chk(delta, x.dtype, "delta")
        xo = torch.empty_like(x)
    else:
        xo = x
    _call("s6d_add_layernorm_" + sfx, _ptr(x), _ptr(delta) if delta is not None else _vp(0), _ptr(gamma), _ptr(beta),
          ctypes.c_float(eps), ctypes.c_long(rows), int(C), _ptr(xo) if delta is not None else _vp(0), _ptr(y),
          _stream())
    return xo, y


def geo_embedding(idx4, Wd, bd, Wa, ba, div_term, out_dtype=torch.float32, split=None):
    """idx4 (...,4) f32 [d_idx, a_idx x3] -> (...,256) geometric structure embedding, stored in f32 or (out_dtype=torch.float16) in
    IEEE half for the RPE attention core to stream.  split = (Wd_hilo, Wa_hilo): the weights already split into bf16 hi / lo parts,
    each a (2, 256, 256) bf16 tensor [hi | lo] (the caller caches them per weight version): same bits, half the kernel's vector work."""
    for a, nm in ((idx4, "idx4"), (Wd, "Wd"), (bd, "bd"), (Wa, "Wa"), (ba, "ba"), (div_term, "div_term")):
        _chk(a, torch.float32, nm)
    if out_dtype not in (torch.float32, torch.float16):
        raise RuntimeError("the embedding is stored in float32 or float16")
    NP = idx4.numel() // 4
    out = torch.empty(*idx4.shape[:-1], Wd.shape[0], dtype=out_dtype, device=idx4.device)
    if split is not None and have("geo_embedding_split"):
        for a, nm in ((split[0], "Wd_hilo"), (split[1], "Wa_hilo")):
            _chk(a, torch.bfloat16, nm, 3)
        _call("s6d_geo_embedding_split", _ptr(idx4), ctypes.c_long(NP), _ptr(split[0]), _ptr(bd), _ptr(split[1]), _ptr(ba), _ptr(div_term),
              int(Wd.shape[0]), int(idx4.shape[-1] - 1), _ptr(out), 1 if out_dtype == torch.float16 else 0, _stream())
        return out
    _call("s6d_geo_embedding_f32" if out_dtype == torch.float32 else "s6d_geo_embedding_f16", _ptr(idx4), ctypes.c_long(NP), _ptr(Wd),
          _ptr(bd), _ptr(Wa), _ptr(ba), _ptr(div_term), int(Wd.shape[0]), int(idx4.shape[-1] - 1), _ptr(out), _stream())
    return out


def fine_assign(atten, pts2):
    """atten (B,M1,M2) f32, pts2 (B,M2-1,3) f32 -> pred (B,M1-1,3), wsum (B,M1-1), w1 (B,M1-1)."""
    _chk(atten, torch.float32, "atten", 3)
    _chk(pts2, torch.float32, "pts2", 3)
    B, M1, M2 = atten.shape
    dev = atten.device
    fn = _lib.lib().s6d_fine_assign_workspace_bytes
    fn.restype = ctypes.c_long
    ws = torch.empty(int(fn(B, M1, M2)), dtype=torch.uint8, device=dev)
    pred = torch.empty(B, M1 - 1, 3, dtype=torch.float32, device=dev)
    wsum = torch.empty(B, M1 - 1, dtype=torch.float32, device=dev)
    w1 = torch.empty(B, M1 - 1, dtype=torch.float32, device=dev)
    _call("s6d_fine_assign_f32", _ptr(atten), _ptr(pts2), B, M1, M2, _ptr(ws), _ptr(pred), _ptr(wsum), _ptr(w1),
          _stream())
    return pred, wsum, w1


def fine_match(f1, f2, pts2, temp):
    """f1 (B,M1,256), f2 (B,M2,256) f32 (background token at row 0), pts2 (B,M2-1,3) -> pred (B,M1-1,3), wsum, w1 (B,M1-1):
    feature similarity + dual softmax + labels + normalised assignment without the (B,M1,M2) matrix."""
    for t, nm in ((f1, "f1"), (f2, "f2"), (pts2, "pts2")):
        _chk(t, torch.float32, nm, 3)
    B, M1, C = f1.shape
    M2 = f2.shape[1]
    if f2.shape[0] != B or f2.shape[2] != C or tuple(pts2.shape) != (B, M2 - 1, 3):
        raise ValueError(f"fine_match: shapes {tuple(f1.shape)}, {tuple(f2.shape)}, {tuple(pts2.shape)}")
    dev = f1.device
    fn = _lib.lib().s6d_fine_match_workspace_bytes
    fn.restype = ctypes.c_long
    ws = torch.empty(int(fn(B, M1, M2)), dtype=torch.uint8, device=dev)
    pred = torch.empty(B, M1 - 1, 3, dtype=torch.float32, device=dev)
    wsum = torch.empty(B, M1 - 1, dtype=torch.float32, device=dev)
    w1 = torch.empty(B, M1 - 1, dtype=torch.float32, device=dev)
    _call("s6d_fine_match_f32", _ptr(f1), _ptr(f2), _ptr(pts2), B, M1, M2, C, ctypes.c_float(1.0 / temp), _ptr(ws),
          _ptr(pred), _ptr(wsum), _ptr(w1), _stream())
    return pred, wsum, w1


def pe_group_mlp(pts, idx, W0, b0, W1, b1, W2, b2):
    """pts (B,N,3) f32, idx (B,N,ns) i32, folded MLP weights -> (B,N,128) f32 (max over neighbours)."""
    _chk(pts, torch.float32, "pts", 3)
    _chk(idx, torch.int32, "idx", 3)
    ws = [w.contiguous() for w in (W0, b0, W1, b1, W2, b2)]
    for w, shp in zip(ws, ((32, 6), (32,), (64, 32), (64,), (128, 64), (128,))):
        _chk(w, torch.float32, "mlp weight")
        if tuple(w.shape) != shp:
            raise RuntimeError(f"SharedMLP [6,32,64,128] expected, got weight of shape {tuple(w.shape)}")
    B, N, _ = pts.shape
    out = torch.empty(B, N, 128, dtype=torch.float32, device=pts.device)
    _call("s6d_pe_group_mlp_f32", _ptr(pts), _ptr(idx), B, N, int(idx.shape[2]), *[_ptr(w) for w in ws], _ptr(out),
          _stream())
    return out


def mha(q, k, v, scale):
    """q (B,N,256), k/v (B,M,256) f32 (column blocks of one projection output are taken as they are) -> (B,N,256): 4-head softmax
    attention, one wavefront per query row."""
    (q, ldq), (k, ldk), (v, ldv) = _rows3(q, "q"), _rows3(k, "k"), _rows3(v, "v")
    B, N, C = q.shape
    out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
    _call("s6d_mha_strided_f32", _ptr(q), ctypes.c_long(ldq), _ptr(k), ctypes.c_long(ldk), _ptr(v), ctypes.c_long(ldv), B, N,
          int(k.shape[1]), C, 4, ctypes.c_float(scale), _ptr(out), _stream())
    return out


def linear_attn_focus(x, inv_scale, power):
    """x (...,256) f32, inv_scale (256) -> focused feature map, same shape."""
    x = x.contiguous()
    _chk(x, torch.float32, "x")
    inv_scale = inv_scale.reshape(-1).contiguous()
    _chk(inv_scale, torch.float32, "inv_scale", 1)
    y = torch.empty_like(x)
    _call("s6d_linear_attn_focus_f32", _ptr(x), _ptr(inv_scale), ctypes.c_long(x.numel() // x.shape[-1]),
          int(x.shape[-1]), int(power), _ptr(y), _stream())
    return y


def linear_attention(q_proj, inv_scale, power, k_focused, v):
    """Focused linear attention behind the projections: q_proj (B,I,256) f32 raw proj_q output, k_focused (B,J,256) f32 (rows may be
    strided), v (B,J,256) f32 (rows may be strided), inv_scale (256) -> (B,I,256) f32."""
    for t, nm in ((q_proj, "q_proj"), (k_focused, "k_focused"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3:
            raise RuntimeError(f"{nm} must be a 3-d float32 CUDA tensor")
    q_proj = q_proj.contiguous()
    B, I, C = q_proj.shape
    J = k_focused.shape[1]
    for t, nm in ((k_focused, "k_focused"), (v, "v")):
        if t.shape != (B, J, C) or t.stride(2) != 1 or (B > 1 and t.stride(0) != J * t.stride(1)):     # (B == 1: stride(0) is free)
            raise ValueError(f"{nm} must be (B, J, C) with unit channel stride and batch stride J * row stride; got {tuple(t.shape)} "
                             f"strides {tuple(t.stride())}")
    inv_scale = inv_scale.reshape(-1).contiguous()
    _chk(inv_scale, torch.float32, "inv_scale", 1)
    fn = _lib.lib().s6d_linear_attention_workspace_floats
    fn.restype = ctypes.c_long
    ws = torch.empty(int(fn(int(B))), dtype=torch.float32, device=q_proj.device)
    out = torch.empty_like(q_proj)
    _call("s6d_linear_attention_f32", _ptr(q_proj), _ptr(inv_scale), int(power), _ptr(k_focused), ctypes.c_long(k_focused.stride(1)),
          _ptr(v), ctypes.c_long(v.stride(1)), int(B), int(I), int(J), int(C), _ptr(ws), _ptr(out), _stream())
    return out


# ------------------------------------------------------------------ ISM scoring
def pairwise_cosine(query, ref):
    """(P,C), (R,C) f32 -> (P,R) clamp(cos,0,1)."""
    _chk(query, torch.float32, "query", 2)
    _chk(ref, torch.float32, "ref", 2)
    P, C = query.shape
    R = ref.shape[0]
    out = torch.empty(P, R, dtype=torch.float32, device=query.device)
    _call("s6d_pairwise_cosine_f32", _ptr(query), _ptr(ref), P, R, C, _ptr(out), _stream())
    return out


def semantic_select(scores, topk):
    """(P,O,T) f32 -> best_score (P) f32, best_obj (P) i32, best_tmpl (P) i32."""
    _chk(scores, torch.float32, "scores", 3)
    P, O, T = scores.shape
    dev = scores.device
    bs = torch.empty(P, dtype=torch.float32, device=dev)
    bo = torch.empty(P, dtype=torch.int32, device=dev)
    bt = torch.empty(P, dtype=torch.int32, device=dev)
    _call("s6d_semantic_select_f32", _ptr(scores), P, O, T, int(topk), _ptr(bs), _ptr(bo), _ptr(bt), _stream())
    return bs, bo, bt


def patch_scores(query, refstore, obj, tmpl, thred, sel=None):
    """query (S,N1,C), refstore (O,T,N2,C) f32, obj/tmpl (S) i32 -> appe (S), ratio (S).
    sel (S) i32: query is the UN-gathered (P,N1,C) tensor of every proposal and row s reads query[sel[s]] (no gathered copy)."""
    _chk(query, torch.float32, "query", 3)
    _chk(refstore, torch.float32, "refstore", 4)
    _chk(obj, torch.int32, "obj", 1)
    _chk(tmpl, torch.int32, "tmpl", 1)
    S, N1, C = query.shape
    if sel is not None:
        _chk(sel, torch.int32, "sel", 1)
        S = sel.shape[0]
        if obj.shape[0] != S or tmpl.shape[0] != S:
            raise ValueError("patch_scores: obj / tmpl / sel must have one entry per selected proposal")
    _, T, N2, _ = refstore.shape
    fn = _lib.lib().s6d_patch_scores_workspace_floats
    fn.restype = ctypes.c_long
    ws = torch.empty(max(int(fn(S, N1, N2)), 1), dtype=torch.float32, device=query.device)
    appe = torch.empty(S, dtype=torch.float32, device=query.device)
    ratio = torch.empty(S, dtype=torch.float32, device=query.device)
    _call("s6d_patch_scores_sel_f32", _ptr(query), _ptr(sel) if sel is not None else _vp(0), _ptr(refstore), _ptr(obj), _ptr(tmpl),
          S, N1, N2, C, T, ctypes.c_float(thred), _ptr(ws), _ptr(appe), _ptr(ratio), _stream())
    return appe, ratio


def masked_depth_mean(masks, depth, K, depth_scale, frame=None, sel=None):
    """sel (S) i32: masks is the UN-gathered (P,H,W) tensor of every proposal and mask s of the call is masks[sel[s]].
    masks (S,H,W) f32, depth (H,W) f32, K 3x3 (any float dtype, host or device) -> (S,3) f32.  The camera matrix goes to
    the device as float64 (the reference's dtype) and is read there: no host copy of a device K and no cache keyed by
    an address (a new frame's K may land on the old one's).  Several frames in one launch: depth (F,H,W), K (F,3,3),
    frame (S) i32 = the frame of every mask."""
    _chk(masks, torch.float32, "masks", 3)
    S, H, W = masks.shape
    if sel is not None:
        _chk(sel, torch.int32, "sel", 1)
        S = sel.shape[0]
    if frame is None:
        _chk(depth, torch.float32, "depth", 2)
        if tuple(K.shape) != (3, 3):
            raise ValueError(f"K must be 3x3, got {tuple(K.shape)}")
    else:
        _chk(depth, torch.float32, "depth", 3)
        _chk(frame, torch.int32, "frame", 1)
        if K.dim() != 3 or tuple(K.shape[1:]) != (3, 3) or K.shape[0] != depth.shape[0] or frame.shape[0] != S:
            raise ValueError(f"batched call: depth (F,H,W), K (F,3,3), frame (S); got {tuple(depth.shape)}, {tuple(K.shape)}, {tuple(frame.shape)}")
    if frame is not None and policy.current().debug and S:          # costs a host round trip: debug runs only
        lo, hi = int(frame.min()), int(frame.max())
        if lo < 0 or hi >= depth.shape[0]:
            raise ValueError(f"frame indices span [{lo}, {hi}] but there are {depth.shape[0]} frames")
    Kd = K.detach().to(device=masks.device, dtype=torch.float64).contiguous()
    out = torch.empty(S, 3, dtype=torch.float32, device=masks.device)
    fn = _lib.lib().s6d_masked_depth_mean_workspace_bytes
    fn.restype = ctypes.c_long
    ws = torch.empty(max(int(fn(S, H, W)), 8), dtype=torch.uint8, device=masks.device)
    _call("s6d_masked_depth_mean_sel_f32", _ptr(masks), _ptr(sel) if sel is not None else _vp(0), _ptr(depth),
          _ptr(frame) if frame is not None else _vp(0), S, H, W, ctypes.c_float(depth_scale), _ptr(Kd), _ptr(ws), _ptr(out), _stream())
    return out


def project_bbox(pointcloud, poses, obj, tmpl, trans, K, H, W, frame=None):
    """-> uv (S,N,2) i32, bbox (S,4) i32 = (min u, min v, max u, max v).  K (3,3) f32, or (F,3,3) with frame (S) i32."""
    for a, nm, nd in ((pointcloud, "pointcloud", 3), (poses, "poses", 3), (trans, "trans", 2), (K, "K", 2 if frame is None else 3)):
        _chk(a, torch.float32, nm, nd)
    _chk(obj, torch.int32, "obj", 1)
    _chk(tmpl, torch.int32, "tmpl", 1)
    if frame is not None:
        _chk(frame, torch.int32, "frame", 1)
    S, N = obj.shape[0], pointcloud.shape[1]
    uv = torch.empty(S, N, 2, dtype=torch.int32, device=trans.device)
    bbox = torch.empty(S, 4, dtype=torch.int32, device=trans.device)
    _call("s6d_project_bbox_frames_f32", _ptr(pointcloud), _ptr(poses), _ptr(obj), _ptr(tmpl), _ptr(trans), _ptr(K),
          _ptr(frame) if frame is not None else _vp(0), S, N, int(H), int(W), _ptr(uv), _ptr(bbox), _stream())
    return uv, bbox


# ------------------------------------------------------------------ fused-op registry
# Names of fused gfx950 ops the loaded library exports.  Product modules ask ``have(name)``
# and otherwise express the same math with library GEMMs on the device (never on the CPU).
_FUSED = {}


def have(name):
    if name not in _FUSED:
        sym = {"rpe_attention": "s6d_rpe_attention_f32", "rpe_attention_packed": "s6d_rpe_attention_packed_f32", "geo_embedding": "s6d_geo_embedding_f32", "geo_embedding_f16": "s6d_geo_embedding_f16", "geo_embedding_split": "s6d_geo_embedding_split",
               "fine_assign": "s6d_fine_assign_f32", "fine_match": "s6d_fine_match_f32", "pem_pre": "s6d_pem_compact_cloud_f32", "coarse_sample": "s6d_coarse_sample_f32", "upsample_gather": "s6d_upsample_gather_f32",
               "min_dist": "s6d_min_dist_f32", "rot_from_h": "s6d_rot_from_h_f32", "weighted_procrustes": "s6d_weighted_procrustes_f32", "add_layernorm": "s6d_add_layernorm_bf16", "gemm_bf16": "s6d_gemm_bf16", "gemm_bf16_res": "s6d_gemm_bf16_res", "gemm_bf16_lnfold": "s6d_gemm_bf16_lnfold", "gemm_f16": "s6d_gemm_f16", "gemm_fp8": "s6d_gemm_fp8", "layernorm_fp8": "s6d_layernorm_fp8", "layernorm_f32out": "s6d_layernorm_bf16_f32", "linear_f32": "s6d_linear_f32", "attn_output_chain": "s6d_attn_output_chain_f32", "win_attention": "s6d_win_attention_layout_bf16",
               "glb_attention": "s6d_glb_attention_bf16", "pairwise_cosine": "s6d_pairwise_cosine_f32",
               "patch_scores": "s6d_patch_scores_sel_f32", "pose_hypotheses": "s6d_pose_hypotheses_f32",
               "pe_group": "s6d_pe_group_mlp_f32", "masked_depth_mean": "s6d_masked_depth_mean_sel_f32",
               "semantic_select": "s6d_semantic_select_f32", "seq_attention": "s6d_seq_attention_bf16", "sam_preprocess": "s6d_sam_preprocess_f32", "im2col3x3": "s6d_im2col3x3_b16", "nonfinite_rows": "s6d_nonfinite_rows_f32", "patchify": "s6d_patchify_b16", "crop_resize_pad": "s6d_crop_resize_pad_f32", "samdec_img2tok": "s6d_samdec_img2tok_bf16", "samdec_img2tok_raw": "s6d_samdec_img2tok_raw_bf16", "samdec_tok2img": "s6d_samdec_tok2img_f32", "samdec_tok2img_raw": "s6d_samdec_tok2img_raw_bf16", "sam_mask_post": "s6d_sam_mask_post_sel_f32", "gemm_fp8_mx": "s6d_gemm_fp8_mxa", "nms": "s6d_nms_f32", "samdec_upscale_heads": "s6d_samdec_upscale_heads_bf16", "samdec_tokens": "s6d_samdec_tokens_post_bf16", "samdec_token_folds": "s6d_samdec_tokens_post_bf16", "mha": "s6d_mha_f32",
               "linear_attn_focus": "s6d_linear_attn_focus_f32", "linear_attention": "s6d_linear_attention_f32", "project_bbox": "s6d_project_bbox_frames_f32"}.get(name)
        _FUSED[name] = sym is not None and hasattr(_lib.lib(), sym)
    # (policy.disable_fused: kernel names the modules must not use -- the tests' way of forcing the library statement)
    if policy.current().disable_fused and name in policy.current().disable_fused.split(","):
        return False
    return _FUSED[name]
