"""Pose solvers of the PEM matching heads, MI355X build.

Re-derivation of Pose_Estimation_Model/utils/model_utils.py
(compute_coarse_Rt :187-246, compute_fine_Rt :250-283, weighted_procrustes :287-363):
same quantities, different execution -- hypothesis generation, the 3x3 SVDs and the
transform + nearest-model-point scans run as fused gfx950 kernels when the library exports
them (ops.have), and as device-side library ops otherwise.
"""
import torch

from .. import ops
from .. import policy


def soft_assignment(atten):
    """softmax over rows x softmax over columns, background row/column masking
    (model_utils.py:203-212 / 262-266).  Returns (masked score[:,1:,1:], w1, w2)."""
    score = torch.softmax(atten, dim=2) * torch.softmax(atten, dim=1)
    w1 = (score[:, 1:, :].argmax(dim=2) > 0).float()
    w2 = (score[:, :, 1:].argmax(dim=1) > 0).float()
    return score[:, 1:, 1:] * w1.unsqueeze(2) * w2.unsqueeze(1), w1, w2


def rotation_from_H(H):
    """R = V diag(1,1,det(V U^T)) U^T for H = U S V^T (model_utils.py:343-347)."""
    if policy.guard("pem.rot_from_h", cuda=H.is_cuda, have=ops.have("rot_from_h")):
        return ops.rot_from_h(H.contiguous())
    U, _, Vh = torch.linalg.svd(H.double())
    V = Vh.transpose(-1, -2)
    d = torch.sign(torch.det(V @ U.transpose(-1, -2)))
    D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), d], dim=-1))
    return (V @ D @ U.transpose(-1, -2)).to(H.dtype)


def weighted_procrustes(src, ref, weights=None, weight_thresh=0.0, eps=1e-5):
    if weights is None:
        weights = torch.ones_like(src[..., 0])
    if policy.guard("pem.weighted_procrustes", cuda=src.is_cuda, have=ops.have("weighted_procrustes"), batched=src.dim() == 3,
                    f32=src.dtype == torch.float32):
        # one launch, fixed summation order per instance: the pose of an instance does not depend on the batch it is in
        return ops.weighted_procrustes(src.contiguous(), ref.contiguous(), weights.contiguous(), weight_thresh, eps)
    weights = torch.where(weights < weight_thresh, torch.zeros_like(weights), weights)
    w = (weights / (weights.sum(dim=-1, keepdim=True) + eps)).unsqueeze(-1)
    sc = (src * w).sum(dim=-2, keepdim=True)
    rc = (ref * w).sum(dim=-2, keepdim=True)
    H = (src - sc).transpose(-1, -2) @ (w * (ref - rc))
    R = rotation_from_H(H)
    t = (rc.transpose(-1, -2) - R @ sc.transpose(-1, -2)).squeeze(-1)
    return R, t


def min_dist_to_model(pts, R, t, model):
    """For every pose p and point n: min_m || (pts[n] - t_p) R_p - model[m] ||.
    pts (B,N,3), R (B,P,3,3), t (B,P,3), model (B,Nm,3) -> (B,P,N)."""
    if policy.guard("pem.min_dist", cuda=pts.is_cuda, have=ops.have("min_dist")):
        return ops.min_dist(pts.contiguous(), R.contiguous(), t.contiguous(), model.contiguous())
    tp = (pts.unsqueeze(1) - t.unsqueeze(2)) @ R                       # (B,P,N,3)
    out = []
    for chunk in tp.split(64, dim=1):                                  # bound the (P,N,Nm) intermediate
        d = chunk.unsqueeze(3) - model.unsqueeze(1).unsqueeze(1)       # (B,p,N,Nm,3)
        out.append((d * d).sum(-1).min(dim=3)[0].sqrt())
    return torch.cat(out, dim=1)


def coarse_Rt(atten, pts1, pts2, model_pts, rand_u, n1=6000, n2=300):
    """compute_coarse_Rt.  rand_u (B, 3*n1) are the uniform samples (drawn by the caller)."""
    B, N1, _ = pts1.shape
    N2 = pts2.shape[1]
    if policy.guard("pem.compute_coarse_Rt", cuda=atten.is_cuda, have=ops.have("coarse_sample"), f32=atten.dtype == torch.float32,
                    N2_le_255=N2 + 1 <= 256, lds=N1 * N2 * 4 <= 152 * 1024, n1_le_16384=n1 <= 16384):
        # five launches: sampling head (dual softmax, labels, ^1.5, prefix sums, search) -> hypotheses -> the n2 smallest residuals
        # -> nearest-model-point scan -> scored arg-max
        pair, w1 = ops.coarse_sample(atten.contiguous(), rand_u.contiguous())
        Rs, ts, dis = ops.pose_hypotheses(pts1.contiguous(), pts2.contiguous(), pair)
        Rk, tk, _ = ops.smallest_k(dis, Rs, ts, n2)
        dmin = ops.min_dist(pts1.contiguous(), Rk, tk, model_pts.contiguous())
        return ops.hypothesis_select(dmin, w1, Rk, tk)
    score, w1, _ = soft_assignment(atten)
    score = score.reshape(B, N1 * N2) ** 1.5
    cum = torch.cumsum(score, dim=1)
    cum = cum / (cum[:, -1:].contiguous() + 1e-8)
    pair = torch.searchsorted(cum, rand_u.contiguous())
    if policy.guard("pem.pose_hypotheses", cuda=pts1.is_cuda, have=ops.have("pose_hypotheses")):
        Rs, ts, dis = ops.pose_hypotheses(pts1.contiguous(), pts2.contiguous(), pair.int().contiguous())
    else:
        i1 = torch.clamp(pair.div(N2, rounding_mode="floor"), max=N1 - 1)
        i2 = torch.clamp(pair % N2, max=N2 - 1)
        p1 = torch.gather(pts1, 1, i1.unsqueeze(2).expand(-1, -1, 3)).view(B, n1, 3, 3)
        p2 = torch.gather(pts2, 1, i2.unsqueeze(2).expand(-1, -1, 3)).view(B, n1, 3, 3)
        Rs, ts = weighted_procrustes(p2, p1, None, weight_thresh=0.5)
        dis = torch.norm((p1 - ts.unsqueeze(2)) @ Rs - p2, dim=3).mean(dim=2)
    top = torch.topk(dis, n2, dim=1, largest=False)[1]
    Rs = torch.gather(Rs, 1, top.view(B, n2, 1, 1).expand(-1, -1, 3, 3))
    ts = torch.gather(ts, 1, top.view(B, n2, 1).expand(-1, -1, 3))
    dmin = min_dist_to_model(pts1, Rs, ts, model_pts)                  # (B,n2,N1)
    sc = w1.sum(dim=1, keepdim=True) / ((dmin * w1.unsqueeze(1)).sum(dim=2) + 1e-8)
    best = sc.argmax(dim=1)
    ar = torch.arange(B, device=pts1.device)
    return Rs[ar, best], ts[ar, best]


def fine_Rt(atten, pts1, pts2, model_pts, dis_thres=0.15, feats=None):
    """compute_fine_Rt.  feats = (f1, f2, temp): the out_proj features the similarity is made of -- with them the fused kernel
    forms similarity tiles on the fly and ``atten`` (the (B,2049,2049) matrix) may be None."""
    if feats is not None and policy.guard("pem.compute_fine_Rt", cuda=feats[0].is_cuda, have=ops.have("fine_match"),
                                          C256=feats[0].shape[2] == 256, f32=feats[0].dtype == torch.float32):
        pred, wsum, w1 = ops.fine_match(feats[0].contiguous(), feats[1].contiguous(), pts2.contiguous(), float(feats[2]))
    elif atten is None:
        raise ValueError("fine_Rt needs the similarity matrix when the fused similarity kernel is not available")
    elif policy.guard("pem.compute_fine_Rt.assign", cuda=atten.is_cuda, have=ops.have("fine_assign"), cols=atten.shape[2] <= 2112,
                      background_column=pts2.shape[1] == atten.shape[2] - 1):
        pred, wsum, w1 = ops.fine_assign(atten.contiguous(), pts2.contiguous())
    else:
        amat, w1, _ = soft_assignment(atten)
        wsum = amat.sum(dim=2)
        pred = (amat / (wsum.unsqueeze(2) + 1e-6)) @ pts2
    R, t = weighted_procrustes(pred, pts1, wsum, weight_thresh=0.0)
    dis = min_dist_to_model(pts1, R.unsqueeze(1), t.unsqueeze(1), model_pts).squeeze(1)
    sc = ((dis < dis_thres).float() * w1).sum(dim=1) / (w1.sum(dim=1) + 1e-8)
    return R, t, sc * w1.mean(dim=1)
