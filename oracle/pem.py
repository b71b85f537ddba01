"""CPU restatement of the reference Pose Estimation Model forward -- TEST INFRASTRUCTURE ONLY.

Functional torch-fp32 (CPU) restatement of
``Pose_Estimation_Model/model/{pose_estimation_model,feature_extraction,transformer,
coarse_point_matching,fine_point_matching}.py`` and ``utils/model_utils.py`` of the
reference.  Weights are a flat ``{state_dict key: tensor}`` mapping using the reference's
own key names, so the same mapping drives the reference modules (golden generation), this
oracle and the product modules.  Pinned against the reference itself: tests/golden/pem_*.npz
are produced by oracle/gen_golden.py from the reference modules imported unmodified, and
tests/test_oracle_golden.py checks this file against them.

Exception: the timm ViT used by feature_extraction.py is un-vendored and unpinned --
PARITY UNPINNED at that boundary (see oracle/timm_standin.py).

The only deliberate interface difference from the reference: the 18000 uniform samples
that compute_coarse_Rt draws with torch.rand (model_utils.py:219) are an INPUT (``rand_u``)
so that results are comparable across devices.
"""
import math

import torch
import torch.nn.functional as F

from . import pn2

CFG = dict(coarse_npoint=196, fine_npoint=2048, sigma_d=0.2, sigma_a=15.0, angle_k=3,
           hidden=256, heads=4, nblock=3, temp=0.1, nproposal1=6000, nproposal2=300,
           pe_r1=0.1, pe_r2=0.2, pe_ns1=32, pe_ns2=64, focusing_factor=3,
           vit_depth=12, vit_heads=12, vit_dim=768)


# ----------------------------------------------------------------------------- helpers
def lin(W, p, x):
    return F.linear(x, W[p + ".weight"], W[p + ".bias"])


def lnorm(W, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"], eps)


def pairwise_sqdist(x, y):
    """model_utils.py:84-111 (normalized=False, channel_first=False)."""
    xy = x @ y.transpose(-1, -2)
    x2 = (x ** 2).sum(-1).unsqueeze(-1)
    y2 = (y ** 2).sum(-1).unsqueeze(-2)
    return (x2 - 2 * xy + y2).clamp(min=0.0)


# ----------------------------------------------------------------------------- ViT (a10-a12)
def vit_pyramid(W, p, rgb, depth=12, heads=12):
    """feature_extraction.py:17-35 on top of timm's VisionTransformer (restated, see
    timm_standin.py).  Returns the 4 normalised taps in block order [2,5,8,11]."""
    x = F.conv2d(rgb, W[p + ".patch_embed.proj.weight"], W[p + ".patch_embed.proj.bias"], stride=16)
    x = x.flatten(2).transpose(1, 2)                                   # (B,196,D)
    B, _, D = x.shape
    x = torch.cat([W[p + ".cls_token"].expand(B, -1, -1), x], 1) + W[p + ".pos_embed"]
    n = depth // 4
    taps = {depth - 1, depth - n - 1, depth - 2 * n - 1, depth - 3 * n - 1}
    hd = D // heads
    out = []
    for i in range(depth):
        bp = f"{p}.blocks.{i}"
        h = lnorm(W, bp + ".norm1", x, 1e-6)
        qkv = lin(W, bp + ".attn.qkv", h).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
        h = (a @ v).transpose(1, 2).reshape(B, -1, D)
        x = x + lin(W, bp + ".attn.proj", h)
        h = lnorm(W, bp + ".norm2", x, 1e-6)
        x = x + lin(W, bp + ".mlp.fc2", F.gelu(lin(W, bp + ".mlp.fc1", h)))
        if i in taps:
            out.append(lnorm(W, p + ".norm", x, 1e-6))
    return out


def vit_ae_featmap(W, p, rgb):
    """feature_extraction.py:98-117, up_type == 'linear', use_pyramid_feat.
    (B,3,224,224) -> (B,256,224,224)."""
    B, _, H, Wd = rgb.shape
    taps = [t[:, 1:, :] for t in vit_pyramid(W, p + ".vit", rgb)]
    x = lin(W, p + ".output_upscaling", torch.cat(taps, 2))           # (B,196,4096)
    x = x.reshape(B, 14, 14, 4, 4, -1).permute(0, 5, 1, 3, 2, 4).reshape(B, -1, 56, 56)
    return F.interpolate(x, (H, Wd), mode="bilinear", align_corners=False)


def chosen_pixel_feats(fmap, choose):
    """model_utils.py:69-81.  fmap (B,C,H,W), choose (B,n) int64 -> (B,n,C)."""
    B, C = fmap.shape[:2]
    flat = fmap.reshape(B, C, -1)
    return torch.gather(flat, 2, choose.unsqueeze(1).expand(-1, C, -1)).transpose(1, 2).contiguous()


def feature_extraction(W, ep):
    """ViTEncoder.forward, eval branch with dense_po/dense_fo given (feature_extraction.py:128-146)."""
    p = "feature_extraction.rgb_net"
    dense_fm = chosen_pixel_feats(vit_ae_featmap(W, p, ep["rgb"]), ep["rgb_choose"])
    dense_po = ep["dense_po"].clone()
    dense_fo = ep["dense_fo"].clone()
    radius = torch.norm(dense_po, dim=2).max(1)[0]
    s = radius.reshape(-1, 1, 1) + 1e-6
    return ep["pts"] / s, dense_fm, dense_po / s, dense_fo, radius


def get_obj_feats(W, tem_rgb_list, tem_pts_list, tem_choose_list, npoint=2048):
    """ViTEncoder.get_obj_feats (feature_extraction.py:170-181): template onboarding."""
    p = "feature_extraction.rgb_net"
    feats = [chosen_pixel_feats(vit_ae_featmap(W, p, t), c) for t, c in zip(tem_rgb_list, tem_choose_list)]
    return sample_pts_feats(torch.cat(tem_pts_list, 1), torch.cat(feats, 1), npoint)[:2]


# ----------------------------------------------------------------------------- sampling (a13)
def sample_pts_feats(pts, feats, npoint):
    """model_utils.py:53-66 -> (pts (B,n,3), feats (B,n,C), idx (B,n) int32)."""
    idx = pn2.furthest_point_sampling(pts.contiguous(), npoint)
    li = idx.long()
    g = lambda t: torch.gather(t, 1, li.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    return g(pts), g(feats), idx


# ----------------------------------------------------------------------------- geometric embedding (a14)
def sinusoid(x, d_model=256):
    """transformer.py:257-281: interleaved [sin w0, cos w0, sin w1, cos w1, ...]."""
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    om = x.unsqueeze(-1) * div
    return torch.stack([torch.sin(om), torch.cos(om)], -1).reshape(*x.shape, d_model)


def geo_indices(points, sigma_d=0.2, sigma_a=15.0, k=3):
    """transformer.py:303-332 -> d_idx (B,N,N), a_idx (B,N,N,k)."""
    B, N, _ = points.shape
    dist = torch.sqrt(pairwise_sqdist(points, points))
    knn = dist.topk(k=k + 1, dim=2, largest=False)[1][:, :, 1:]                    # (B,N,k)
    knn_pts = torch.gather(points.unsqueeze(1).expand(B, N, N, 3), 2, knn.unsqueeze(3).expand(B, N, k, 3))
    ref = (knn_pts - points.unsqueeze(2)).unsqueeze(2).expand(B, N, N, k, 3)       # nbr(n) - p_n
    anc = (points.unsqueeze(1) - points.unsqueeze(2)).unsqueeze(3).expand(B, N, N, k, 3)  # p_m - p_n
    sin_v = torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1)
    cos_v = (ref * anc).sum(-1)
    return dist / sigma_d, torch.atan2(sin_v, cos_v) * (180.0 / (sigma_a * math.pi))


def geo_embedding(W, points, p="geo_embedding"):
    """GeometricStructureEmbedding.forward (transformer.py:334-349), reduction 'max'."""
    d_idx, a_idx = geo_indices(points)
    d = lin(W, p + ".proj_d", sinusoid(d_idx))
    a = lin(W, p + ".proj_a", sinusoid(a_idx)).max(dim=3)[0]
    return d + a


# ----------------------------------------------------------------------------- transformer layers (a15, a16)
def _heads(x, h=4):
    B, N, C = x.shape
    return x.reshape(B, N, h, C // h).transpose(1, 2)                  # (B,h,N,c)


def attention_output(W, p, x):
    """AttentionOutput (transformer.py:182-197), ReLU."""
    return lnorm(W, p + ".norm", x + lin(W, p + ".squeeze", F.relu(lin(W, p + ".expand", x))))


def rpe_attention(W, a, x, emb):
    """RPEMultiHeadAttention.forward (transformer.py:368-406), memory == input; returns the
    merged-head hidden states (B,N,C)."""
    q, k, v = (_heads(lin(W, f"{a}.proj_{n}", x)) for n in "qkv")
    B, N, _ = x.shape
    pe = lin(W, a + ".proj_p", emb).reshape(B, N, N, 4, 64).permute(0, 3, 1, 2, 4)   # (B,h,N,M,c)
    s = (torch.einsum("bhnc,bhmc->bhnm", q, k) + torch.einsum("bhnc,bhnmc->bhnm", q, pe)) / 8.0
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, -1)


def rpe_layer(W, p, x, emb):
    """RPETransformerLayer (transformer.py:352-465) with memory == input."""
    h = rpe_attention(W, p + ".attention.attention", x, emb)
    h = lnorm(W, p + ".attention.norm", lin(W, p + ".attention.linear", h) + x)
    return attention_output(W, p + ".output", h)


def cross_layer(W, p, x, mem):
    """TransformerLayer (transformer.py:93-224)."""
    a = p + ".attention.attention"
    q = _heads(lin(W, a + ".proj_q", x))
    k = _heads(lin(W, a + ".proj_k", mem))
    v = _heads(lin(W, a + ".proj_v", mem))
    s = torch.einsum("bhnc,bhmc->bhnm", q, k) / 8.0
    h = (s.softmax(-1) @ v).transpose(1, 2).reshape(x.shape[0], x.shape[1], -1)
    h = lnorm(W, p + ".attention.norm", lin(W, p + ".attention.linear", h) + x)
    return attention_output(W, p + ".output", h)


def geometric_transformer(W, p, f0, e0, f1, e1):
    """GeometricTransformer blocks ['self','cross'], parallel=False (transformer.py:493-513)."""
    f0 = rpe_layer(W, p + ".layers.0", f0, e0)
    f1 = rpe_layer(W, p + ".layers.0", f1, e1)
    f0 = cross_layer(W, p + ".layers.1", f0, f1)
    f1 = cross_layer(W, p + ".layers.1", f1, f0)       # attends to the UPDATED f0
    return f0, f1


def linear_attention(W, p, xq, xkv, focusing=3):
    """LinearAttention (transformer.py:518-564)."""
    q, k, v = lin(W, p + ".proj_q", xq), lin(W, p + ".proj_k", xkv), lin(W, p + ".proj_v", xkv)
    scale = F.softplus(W[p + ".scale"])
    q = (F.relu(q) + 1e-6) / scale
    k = (F.relu(k) + 1e-6) / scale
    qn, kn = q.norm(dim=-1, keepdim=True), k.norm(dim=-1, keepdim=True)
    q, k = q ** focusing, k ** focusing
    q = q / q.norm(dim=-1, keepdim=True) * qn
    k = k / k.norm(dim=-1, keepdim=True) * kn
    B, I, C = q.shape
    sp = lambda t: t.reshape(B, t.shape[1], 4, C // 4).permute(0, 2, 1, 3).reshape(B * 4, t.shape[1], C // 4)
    q, k, v = sp(q), sp(k), sp(v)
    i, j, c, d = q.shape[-2], k.shape[-2], k.shape[-1], v.shape[-1]
    z = 1 / (torch.einsum("bic,bc->bi", q, k.sum(1)) + 1e-6)
    if i * j * (c + d) > c * d * (i + j):
        x = torch.einsum("bic,bcd,bi->bid", q, torch.einsum("bjc,bjd->bcd", k, v), z)
    else:
        x = torch.einsum("bij,bjd,bi->bid", torch.einsum("bic,bjc->bij", q, k), v, z)
    return x.reshape(B, 4, I, C // 4).permute(0, 2, 1, 3).reshape(B, I, C)


def linear_layer(W, p, x, mem):
    """LinearTransformerLayer (transformer.py:567-608)."""
    h = linear_attention(W, p + ".attention.attention", x, mem)
    h = lnorm(W, p + ".attention.norm", lin(W, p + ".attention.linear", h) + x)
    return attention_output(W, p + ".output", h)


def sparse_to_dense(W, p, d0, e0, idx0, d1, e1, idx1):
    """SparseToDenseTransformer.forward (transformer.py:642-673).  NOTE quirk Q1: the FPS
    indices address the tensor that has the bg token prepended, i.e. an off-by-one gather
    (row idx of [bg; dense]) -- reproduced because the trained weights depend on it."""
    def sample(d, idx):
        li = idx.long().unsqueeze(-1).expand(-1, -1, d.shape[-1])
        return torch.cat([d[:, 0:1], torch.gather(d, 1, li)], 1)
    s0, s1 = sample(d0, idx0), sample(d1, idx1)
    s0, s1 = geometric_transformer(W, p + ".sparse_layer", s0, e0, s1, e1)
    n0 = torch.cat([s0[:, 0:1], linear_layer(W, p + ".dense_layer", d0[:, 1:], s0[:, 1:])], 1)
    n1 = torch.cat([s1[:, 0:1], linear_layer(W, p + ".dense_layer", d1[:, 1:], s1[:, 1:])], 1)
    return n0, n1


def feature_similarity(f1, f2, temp=0.1):
    """model_utils.py:114-136, cosine, normalize_feat=True."""
    return F.normalize(f1, p=2, dim=2) @ F.normalize(f2, p=2, dim=2).transpose(1, 2) / temp


# ----------------------------------------------------------------------------- pose solvers (a18, a23)
def weighted_procrustes(src, ref, weights=None, weight_thresh=0.0, eps=1e-5):
    """model_utils.py:287-363."""
    if weights is None:
        weights = torch.ones_like(src[:, :, 0])
    weights = torch.where(weights < weight_thresh, torch.zeros_like(weights), weights)
    w = (weights / (weights.sum(1, keepdim=True) + eps)).unsqueeze(2)
    sc = (src * w).sum(1, keepdim=True)
    rc = (ref * w).sum(1, keepdim=True)
    H = (src - sc).transpose(1, 2) @ (w * (ref - rc))
    U, _, V = torch.svd(H)
    eye = torch.eye(3).unsqueeze(0).repeat(src.shape[0], 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ U.transpose(1, 2)))
    R = V @ eye @ U.transpose(1, 2)
    t = (rc.transpose(1, 2) - R @ sc.transpose(1, 2)).squeeze(2)
    return R, t


def soft_assignment(atten):
    """Shared head of compute_coarse_Rt / compute_fine_Rt (model_utils.py:203-212, 262-266)."""
    score = torch.softmax(atten, 2) * torch.softmax(atten, 1)
    l1 = score[:, 1:, :].max(2)[1]
    l2 = score[:, :, 1:].max(1)[1]
    w1, w2 = (l1 > 0).float(), (l2 > 0).float()
    return score[:, 1:, 1:] * w1.unsqueeze(2) * w2.unsqueeze(1), w1, w2


def coarse_Rt(atten, pts1, pts2, model_pts, rand_u, n1=6000, n2=300):
    """compute_coarse_Rt (model_utils.py:187-246); rand_u (B, 3*n1) replaces torch.rand."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    score, w1, _ = soft_assignment(atten)
    score = score.reshape(B, N1 * N2) ** 1.5
    cum = torch.cumsum(score, 1)
    cum = cum / (cum[:, -1].unsqueeze(1) + 1e-8)
    idx = torch.searchsorted(cum, rand_u)
    i1 = torch.clamp(idx.div(N2, rounding_mode="floor"), max=N1 - 1)
    i2 = torch.clamp(idx % N2, max=N2 - 1)
    p1 = torch.gather(pts1, 1, i1.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1, 3, 3)
    p2 = torch.gather(pts2, 1, i2.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n1, 3, 3)
    Rs, ts = weighted_procrustes(p2, p1, None, weight_thresh=0.5)
    Rs, ts = Rs.reshape(B, n1, 3, 3), ts.reshape(B, n1, 1, 3)
    p1, p2 = p1.reshape(B, n1, 3, 3), p2.reshape(B, n1, 3, 3)
    dis = torch.norm((p1 - ts) @ Rs - p2, dim=3).mean(2)
    top = torch.topk(dis, n2, dim=1, largest=False)[1]
    Rs = torch.gather(Rs, 1, top.reshape(B, n2, 1, 1).expand(-1, -1, 3, 3))
    ts = torch.gather(ts, 1, top.reshape(B, n2, 1, 1).expand(-1, -1, 1, 3))
    tp = ((pts1.unsqueeze(1) - ts) @ Rs).reshape(B * n2, -1, 3)
    mp = model_pts.unsqueeze(1).expand(-1, n2, -1, -1).reshape(B * n2, -1, 3)
    dmin = torch.sqrt(pairwise_sqdist(tp, mp)).min(2)[0].reshape(B, n2, -1)
    sc = w1.unsqueeze(1).sum(2) / ((dmin * w1.unsqueeze(1)).sum(2) + 1e-8)
    best = sc.max(1)[1]
    R = torch.gather(Rs, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 3, 3)).squeeze(1)
    t = torch.gather(ts, 1, best.reshape(B, 1, 1, 1).expand(-1, -1, 1, 3)).squeeze(2).squeeze(1)
    return R, t


def fine_Rt(atten, pts1, pts2, model_pts, dis_thres=0.15):
    """compute_fine_Rt (model_utils.py:250-283)."""
    amat, w1, _ = soft_assignment(atten)
    pred = (amat / (amat.sum(2, keepdim=True) + 1e-6)) @ pts2
    R, t = weighted_procrustes(pred, pts1, amat.sum(2), weight_thresh=0.0)
    tp = (pts1 - t.unsqueeze(1)) @ R
    dis = torch.sqrt(pairwise_sqdist(tp, model_pts)).min(2)[0]
    sc = ((dis < dis_thres).float() * w1).sum(1) / (w1.sum(1) + 1e-8)
    return R, t, sc * w1.mean(1)


# ----------------------------------------------------------------------------- positional encoding (a19)
def shared_mlp(W, p, x):
    """SharedMLP [6,32,64,128] = 3x (1x1 conv, no bias) + BN(eval) + ReLU
    (pytorch_utils.py:25-50, 86-134)."""
    for i in range(3):
        q = f"{p}.layer{i}"
        x = F.conv2d(x, W[q + ".conv.weight"])
        x = F.batch_norm(x, W[q + ".normlayer.bn.running_mean"], W[q + ".normlayer.bn.running_var"],
                         W[q + ".normlayer.bn.weight"], W[q + ".normlayer.bn.bias"], False, 0.0, 1e-5)
        x = F.relu(x)
    return x


def positional_encoding(W, p, pts):
    """PositionalEncoding.forward (fine_point_matching.py:101-125) with pts2 = pts1:
    QueryAndGroup(use_xyz) = [grouped_xyz - centre ; grouped absolute xyz] (6 channels)."""
    pts = pts.contiguous()
    chan = pts.transpose(1, 2).contiguous()
    outs = []
    for mlp, r, ns in (("mlp1", CFG["pe_r1"], CFG["pe_ns1"]), ("mlp2", CFG["pe_r2"], CFG["pe_ns2"])):
        idx = pn2.ball_query(pts, pts, r, ns)
        g = pn2.group_points(chan, idx)                                # (B,3,N,ns)
        x = torch.cat([g - chan.unsqueeze(-1), g], 1)
        outs.append(shared_mlp(W, f"{p}.{mlp}", x).max(dim=3)[0])      # (B,128,N)
    x = torch.cat(outs, 1)
    x = F.conv1d(x, W[p + ".mlp3.conv.weight"], W[p + ".mlp3.conv.bias"])
    return x.transpose(1, 2)


# ----------------------------------------------------------------------------- matching heads (a17, a22)
def coarse_matching(W, p1, f1, g1, p2, f2, g2, radius, model, rand_u, p="coarse_point_matching"):
    B = f1.shape[0]
    bg = W[p + ".bg_token"].expand(B, -1, -1)
    f1 = torch.cat([bg, lin(W, p + ".in_proj", f1)], 1)
    f2 = torch.cat([bg, lin(W, p + ".in_proj", f2)], 1)
    for i in range(CFG["nblock"]):
        f1, f2 = geometric_transformer(W, f"{p}.transformers.{i}", f1, g1, f2, g2)
    atten = feature_similarity(lin(W, p + ".out_proj", f1), lin(W, p + ".out_proj", f2), CFG["temp"])
    R, t = coarse_Rt(atten, p1, p2, model / (radius.reshape(-1, 1, 1) + 1e-6), rand_u,
                     CFG["nproposal1"], CFG["nproposal2"])
    return R, t, atten


def fine_matching(W, p1, f1, g1, idx1, p2, f2, g2, idx2, radius, model, init_R, init_t,
                  p="fine_point_matching"):
    B = p1.shape[0]
    p1_ = (p1 - init_t.unsqueeze(1)) @ init_R
    bg = W[p + ".bg_token"].expand(B, -1, -1)
    f1 = torch.cat([bg, lin(W, p + ".in_proj", f1) + positional_encoding(W, p + ".PE", p1_)], 1)
    f2 = torch.cat([bg, lin(W, p + ".in_proj", f2) + positional_encoding(W, p + ".PE", p2)], 1)
    for i in range(CFG["nblock"]):
        f1, f2 = sparse_to_dense(W, f"{p}.transformers.{i}", f1, g1, idx1, f2, g2, idx2)
    atten = feature_similarity(lin(W, p + ".out_proj", f1), lin(W, p + ".out_proj", f2), CFG["temp"])
    R, t, s = fine_Rt(atten, p1, p2, model / (radius.reshape(-1, 1, 1) + 1e-6))
    return R, t * (radius.reshape(-1, 1) + 1e-6), s, atten


def net_forward(W, ep, rand_u, return_intermediates=False):
    """Net.forward (pose_estimation_model.py:23-53), eval mode."""
    dense_pm, dense_fm, dense_po, dense_fo, radius = feature_extraction(W, ep)
    out = matching_forward(W, dense_pm, dense_fm, dense_po, dense_fo, radius, ep["model"], rand_u,
                           return_intermediates)
    if return_intermediates:
        out["dense_fm"] = dense_fm
    return out


def matching_forward(W, dense_pm, dense_fm, dense_po, dense_fo, radius, model, rand_u,
                     return_intermediates=False):
    """Everything in Net.forward after feature extraction."""
    B = dense_pm.shape[0]
    bg_point = torch.ones(B, 1, 3) * 100
    n = CFG["coarse_npoint"]
    sp_m, sf_m, idx_m = sample_pts_feats(dense_pm, dense_fm, n)
    geo_m = geo_embedding(W, torch.cat([bg_point, sp_m], 1))
    sp_o, sf_o, idx_o = sample_pts_feats(dense_po, dense_fo, n)
    geo_o = geo_embedding(W, torch.cat([bg_point, sp_o], 1))
    init_R, init_t, catt = coarse_matching(W, sp_m, sf_m, geo_m, sp_o, sf_o, geo_o, radius, model, rand_u)
    R, t, s, fatt = fine_matching(W, dense_pm, dense_fm, geo_m, idx_m, dense_po, dense_fo, geo_o, idx_o,
                                  radius, model, init_R, init_t)
    out = dict(init_R=init_R, init_t=init_t, pred_R=R, pred_t=t, pred_pose_score=s)
    if return_intermediates:
        out.update(fps_idx_m=idx_m, fps_idx_o=idx_o, geo_m=geo_m, geo_o=geo_o, coarse_atten=catt,
                   fine_atten=fatt)
    return out
