"""CPU restatement of the ISM proposal-vs-template scoring path -- TEST INFRASTRUCTURE ONLY.

Restates (torch fp32, CPU) ``Instance_Segmentation_Model/model/loss.py`` (PairwiseSimilarity,
MaskedPatch_MatrixSimilarity.compute_straight / compute_visible_ratio), the scoring methods of
``model/detector.py`` (compute_semantic_score :260-296, best_template_pose :198-207,
compute_appearance_score :298-308, compute_geometric_score :310-322,
project_template_to_image :209-232, Calculate_the_query_translation :234-246),
``utils/bbox_utils.py::compute_iou`` (:197-221) and
``utils/trimesh_utils.py::depth_image_to_pointcloud_translate_torch`` (:77-105) as free
functions.  Pinned by tests/golden/ism_*.npz (generated from the reference code itself).
"""
import torch
import torch.nn.functional as F


def pairwise_similarity(query, reference):
    """PairwiseSimilarity.forward (loss.py:27-44): (P,C),(O,T,C) -> (P,O,T).  Quirk Q6: inputs
    are L2-normalised and THEN passed to cosine_similarity (a second normalisation), clamp [0,1]."""
    q = F.normalize(query, dim=-1)[:, None, None, :]
    r = F.normalize(reference, dim=-1)[None]
    return F.cosine_similarity(q, r, dim=-1).clamp(min=0.0, max=1.0)


def semantic_score(query, reference, confidence_thresh=0.2, k=5):
    """compute_semantic_score with aggregation 'avg_5' (ISM_sam.yaml:24) + best_template_pose."""
    scores = pairwise_similarity(query, reference)
    per_obj = torch.topk(scores, k=k, dim=-1)[0].mean(-1)
    per_prop, obj = per_obj.max(-1)
    sel = torch.arange(len(per_prop))[per_prop > confidence_thresh]
    pred_obj = obj[sel]
    best_t = scores[sel].max(-1)[1]
    best_template = torch.gather(best_t, 1, pred_obj[:, None])[:, 0] if len(sel) else best_t.new_zeros(0)
    return sel, pred_obj, per_prop[sel], best_template


def appearance_score(query_patch, ref_patch_store, pred_obj, best_template):
    """compute_appearance_score -> compute_straight (loss.py:52-62)."""
    ref = ref_patch_store[pred_obj, best_template]
    sim = query_patch @ ref.transpose(1, 2)
    factor = torch.count_nonzero(query_patch.sum(-1), dim=-1) + 1e-6
    return (sim.max(-1).values.sum(-1) / factor).clamp(0.0, 1.0), ref


def visible_ratio(query_patch, ref_patch, thred=0.5):
    """compute_visible_ratio (loss.py:64-76)."""
    sim = (query_patch @ ref_patch.transpose(1, 2)).max(1)[0]
    valid = torch.count_nonzero(sim, dim=(1,)) + 1e-6
    return torch.count_nonzero(sim * (sim > thred), dim=(1,)) / valid


def mean_translation(masks, depth, K, depth_scale=1.0):
    """Calculate_the_query_translation + depth_image_to_pointcloud_translate_torch."""
    md = masks * depth[None]
    H, W = depth.shape
    u, v = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
    Z = md * depth_scale / 1000
    X = (u - K[0, 2]) * Z / K[0, 0]
    Y = (v - K[1, 2]) * Z / K[1, 1]
    valid = Z > 0
    n = torch.count_nonzero(valid, dim=(1, 2)) + 1e-8
    t = torch.vstack(((X * valid).sum((1, 2)) / n, (Y * valid).sum((1, 2)) / n, (Z * valid).sum((1, 2)) / n))
    return t.permute(1, 0).to(torch.float32)


def mean_translation_pinned(masks, depth, K, depth_scale=1.0):
    """mean_translation with the three sums taken in the EXPLICIT order of ATen's CPU kernel (oracle/aten_sum.py) instead of
    whatever `torch.sum` does on the host running the tests.  On the x86 hosts seen so far both agree bit for bit
    (tests/test_oracle_golden.py); this is the comparand of the device kernel."""
    import numpy as np

    from . import aten_sum
    md = masks.to(torch.float32) * depth.to(torch.float32)[None]
    S, (H, W) = md.shape[0], depth.shape
    u, v = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
    Z = md * depth_scale / 1000
    K = K.to(torch.float64)
    X = (u - K[0, 2]) * Z / K[0, 0]
    Y = (v - K[1, 2]) * Z / K[1, 1]
    valid = Z > 0
    n = (torch.count_nonzero(valid, dim=(1, 2)) + 1e-8).numpy()
    assert n.dtype == np.float32 and X.dtype == torch.float64 and Z.dtype == torch.float32
    if H * W < 8:
        raise NotImplementedError("below one vector ATen takes a scalar path")
    sx = aten_sum.sum_f64((X * valid).reshape(S, -1).numpy())
    sy = aten_sum.sum_f64((Y * valid).reshape(S, -1).numpy())
    sz = aten_sum.sum_f32((Z * valid).reshape(S, -1).numpy())
    return torch.from_numpy(np.stack(((sx / n).astype(np.float32), (sy / n).astype(np.float32), sz / n), 1))


def project_template(poses, pointcloud, best_template, pred_obj, masks, depth, K, depth_scale=1.0):
    """project_template_to_image (detector.py:209-232) -> int (S,N,2) pixel (u,v)."""
    R = poses[best_template, 0:3, 0:3]
    pc = pointcloud[pred_obj]
    posed = (R @ pc.permute(0, 2, 1)).permute(0, 2, 1) + mean_translation(masks, depth, K, depth_scale)[:, None, :]
    Kf = K[None].repeat(len(pred_obj), 1, 1).to(torch.float32)
    homo = torch.bmm(Kf, posed.permute(0, 2, 1)).permute(0, 2, 1)
    uv = (homo / homo[:, :, -1][:, :, None])[:, :, 0:2].to(torch.int)
    H, W = depth.shape
    uv[:, :, 0].clamp_(min=0, max=W - 1)
    uv[:, :, 1].clamp_(min=0, max=H - 1)
    return uv


def compute_iou(a, b):
    """bbox_utils.compute_iou.  Quirk Q3: ONE empty intersection zeroes the IoU of ALL proposals
    (returns the python scalar 0.0)."""
    tl = torch.max(a[:, 0:2], b[:, 0:2])
    br = torch.min(a[:, 2:4], b[:, 2:4])
    wa, wb, wi = a[:, 2:4] - a[:, 0:2], b[:, 2:4] - b[:, 0:2], br - tl
    if (wi > 0).all():
        ai = wi[:, 0] * wi[:, 1]
        return ai / (wa[:, 0] * wa[:, 1] + wb[:, 0] * wb[:, 1] - ai)
    return 0.0


def geometric_score(image_uv, boxes, query_patch, ref_patch, thred=0.5):
    """compute_geometric_score (detector.py:310-322)."""
    vr = visible_ratio(query_patch, ref_patch, thred)
    xyxy = torch.cat((image_uv.min(1).values, image_uv.max(1).values), -1)
    return compute_iou(xyxy, boxes), vr


def score_frame(inp, confidence_thresh=0.2, visible_thred=0.5):
    """The whole matching stage of run_inference_custom.py:168-200 on one frame."""
    sel, pred_obj, sem, best_t = semantic_score(inp["qry_cls"], inp["ref_cls"], confidence_thresh)
    qp = inp["qry_patch"][sel]
    appe, ref = appearance_score(qp, inp["ref_patch"], pred_obj, best_t)
    uv = project_template(inp["poses"], inp["pointcloud"], best_t, pred_obj, inp["masks"][sel], inp["depth"],
                          inp["K"])
    geo, vr = geometric_score(uv, inp["boxes"][sel], qp, ref, visible_thred)
    final = (sem + appe + geo * vr) / (1 + 1 + vr)
    return dict(sel=sel, pred_obj=pred_obj, semantic=sem, best_template=best_t, appearance=appe,
                iou=geo if torch.is_tensor(geo) else torch.full_like(sem, float(geo)), visible_ratio=vr,
                final=final, image_uv=uv)
