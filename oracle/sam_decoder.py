"""CPU restatement of SAM's prompt encoder + two-way mask decoder -- TEST INFRASTRUCTURE ONLY.

Functional torch-fp32 restatement of ``Instance_Segmentation_Model/segment_anything/modeling/``
  * prompt_encoder.py: PositionEmbeddingRandom (_pe_encoding :183-190, forward :192-204, forward_with_coords :206-214),
    PromptEncoder._embed_points :73-90, _embed_boxes :92-99, forward :128-166 (points / boxes; no mask inputs:
    dense = no_mask_embed broadcast :160-164), get_dense_pe :62-71
  * transformer.py: Attention :185-240, TwoWayAttentionBlock :109-182, TwoWayTransformer.forward :62-106
  * mask_decoder.py: MaskDecoder.predict_masks :106-143, forward :71-104, MLP :148-176; common.py LayerNorm2d
  * sam.py: Sam.postprocess_masks :133-162
Weights: flat {state_dict key: tensor} of the ``Sam`` module (keys ``prompt_encoder.*`` / ``mask_decoder.*``).
Pinned by tests/golden/sam_decoder.npz (reference modules imported unmodified by oracle/gen_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SAM = dict(dim=256, emb=64, img=1024, depth=2, heads=8, mlp=2048, n_multi=3, iou_hidden=256, iou_depth=3)
MINI = dict(dim=64, emb=8, img=128, depth=2, heads=4, mlp=96, n_multi=3, iou_hidden=48, iou_depth=3)


def _pe(G, coords01):
    c = (2 * coords01 - 1) @ G
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(W, cfg):
    """get_dense_pe: (1,C,h,w) encoding of the pixel centres of the embedding grid."""
    h = w = cfg["emb"]
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / h
    x = (grid.cumsum(dim=1) - 0.5) / w
    return _pe(W["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"], torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def prompt_encoder(W, cfg, points=None, labels=None, boxes=None):
    """-> (sparse (B,N,C), dense (B,C,h,w)).  points (B,n,2) in input-image pixels, labels (B,n) in {-1,0,1}."""
    G = W["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    parts, B = [], None
    if points is not None:
        B = points.shape[0]
        p = points + 0.5
        lab = labels
        if boxes is None:                                            # pad with a "not a point"
            p = torch.cat([p, torch.zeros((B, 1, 2))], dim=1)
            lab = torch.cat([lab, -torch.ones((B, 1))], dim=1)
        c = p.clone()
        c[:, :, 0] = c[:, :, 0] / cfg["img"]
        c[:, :, 1] = c[:, :, 1] / cfg["img"]
        e = _pe(G, c.to(torch.float))
        e[lab == -1] = 0.0
        e[lab == -1] += W["prompt_encoder.not_a_point_embed.weight"]
        e[lab == 0] += W["prompt_encoder.point_embeddings.0.weight"]
        e[lab == 1] += W["prompt_encoder.point_embeddings.1.weight"]
        parts.append(e)
    if boxes is not None:
        B = boxes.shape[0]
        c = (boxes + 0.5).reshape(-1, 2, 2).clone()
        c[:, :, 0] = c[:, :, 0] / cfg["img"]
        c[:, :, 1] = c[:, :, 1] / cfg["img"]
        e = _pe(G, c.to(torch.float))
        e[:, 0, :] += W["prompt_encoder.point_embeddings.2.weight"]
        e[:, 1, :] += W["prompt_encoder.point_embeddings.3.weight"]
        parts.append(e)
    B = B or 1
    sparse = torch.cat(parts, dim=1) if parts else torch.empty((B, 0, cfg["dim"]))
    dense = W["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, cfg["emb"], cfg["emb"])
    return sparse, dense


def attention_core(q, k, v, heads):
    """Attention.forward between the input and output projections (transformer.py:218-232): heads split off the channel axis,
    softmax(q k^T / sqrt(d)) v, heads recombined.  q (B,Nq,C), k / v (B,Nk,C) -> (B,Nq,C).  The comparand of the fused
    token<->image attention kernels (tests/test_gpu_sam_decoder.py)."""
    def sep(x):
        b, n, c = x.shape
        return x.reshape(b, n, heads, c // heads).transpose(1, 2)
    q, k, v = sep(q), sep(k), sep(v)
    a = torch.softmax(q @ k.permute(0, 1, 3, 2) / math.sqrt(q.shape[-1]), dim=-1)
    o = (a @ v).transpose(1, 2)
    return o.reshape(o.shape[0], o.shape[1], -1)


def attention(W, p, q, k, v, heads):
    q = F.linear(q, W[p + ".q_proj.weight"], W[p + ".q_proj.bias"])
    k = F.linear(k, W[p + ".k_proj.weight"], W[p + ".k_proj.bias"])
    v = F.linear(v, W[p + ".v_proj.weight"], W[p + ".v_proj.bias"])
    return F.linear(attention_core(q, k, v, heads), W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])


def _ln(W, p, x):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"], 1e-5)


def two_way_transformer(W, p, cfg, image_embedding, image_pe, point_embedding):
    heads = cfg["heads"]
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries, query_pe = point_embedding, point_embedding
    for i in range(cfg["depth"]):
        L = f"{p}.layers.{i}"
        if i == 0:                                                   # skip_first_layer_pe: the output REPLACES queries
            queries = attention(W, L + ".self_attn", queries, queries, queries, heads)
        else:
            q = queries + query_pe
            queries = queries + attention(W, L + ".self_attn", q, q, queries, heads)
        queries = _ln(W, L + ".norm1", queries)
        queries = queries + attention(W, L + ".cross_attn_token_to_image", queries + query_pe, keys + key_pe, keys, heads)
        queries = _ln(W, L + ".norm2", queries)
        m = F.linear(F.relu(F.linear(queries, W[L + ".mlp.lin1.weight"], W[L + ".mlp.lin1.bias"])),
                     W[L + ".mlp.lin2.weight"], W[L + ".mlp.lin2.bias"])
        queries = _ln(W, L + ".norm3", queries + m)
        keys = keys + attention(W, L + ".cross_attn_image_to_token", keys + key_pe, queries + query_pe, queries, heads)
        keys = _ln(W, L + ".norm4", keys)
    queries = queries + attention(W, p + ".final_attn_token_to_image", queries + query_pe, keys + key_pe, keys, heads)
    return _ln(W, p + ".norm_final_attn", queries), keys


def _ln2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def _mlp(W, p, x, n):
    for i in range(n):
        x = F.linear(x, W[f"{p}.layers.{i}.weight"], W[f"{p}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def mask_decoder(W, cfg, image_embeddings, image_pe, sparse, dense, multimask_output=True):
    """MaskDecoder.forward -> (masks (B,3|1,4h,4w) logits, iou_pred (B,3|1))."""
    p = "mask_decoder"
    nt = cfg["n_multi"] + 1
    B = sparse.size(0)
    out_tokens = torch.cat([W[p + ".iou_token.weight"], W[p + ".mask_tokens.weight"]], dim=0)
    tokens = torch.cat((out_tokens.unsqueeze(0).expand(B, -1, -1), sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, B, dim=0) + dense
    pos = torch.repeat_interleave(image_pe, B, dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(W, p + ".transformer", cfg, src, pos, tokens)
    iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:1 + nt, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    u = F.conv_transpose2d(src, W[p + ".output_upscaling.0.weight"], W[p + ".output_upscaling.0.bias"], stride=2)
    u = F.gelu(_ln2d(u, W[p + ".output_upscaling.1.weight"], W[p + ".output_upscaling.1.bias"]))
    u = F.gelu(F.conv_transpose2d(u, W[p + ".output_upscaling.3.weight"], W[p + ".output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([_mlp(W, f"{p}.output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3) for i in range(nt)], dim=1)
    b, c, h, w = u.shape
    masks = (hyper @ u.view(b, c, h * w)).view(b, -1, h, w)
    iou = _mlp(W, p + ".iou_prediction_head", iou_tok, cfg["iou_depth"])
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl], iou[:, sl]


def postprocess_masks(masks, img_size, input_size, original_size):
    """Sam.postprocess_masks: bilinear to the padded square, crop the padding, bilinear to the original size."""
    m = F.interpolate(masks, (img_size, img_size), mode="bilinear", align_corners=False)
    m = m[..., : input_size[0], : input_size[1]]
    return F.interpolate(m, original_size, mode="bilinear", align_corners=False)


def stability_score(masks, mask_threshold, threshold_offset):
    """utils/amg.py calculate_stability_score :156-176: IoU of the masks thresholded at +-offset (one contains the other)."""
    inter = (masks > (mask_threshold + threshold_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (mask_threshold - threshold_offset)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def mask_to_box(masks):
    """utils/amg.py batched_mask_to_box :303-346 for (B,H,W) bool masks: XYXY of the set pixels, [0,0,0,0] if empty."""
    h, w = masks.shape[-2:]
    in_h, _ = torch.max(masks, dim=-1)
    ch = in_h * torch.arange(h, device=masks.device)[None, :]
    bottom, _ = torch.max(ch, dim=-1)
    top, _ = torch.min(ch + h * (~in_h), dim=-1)
    in_w, _ = torch.max(masks, dim=-2)
    cw = in_w * torch.arange(w, device=masks.device)[None, :]
    right, _ = torch.max(cw, dim=-1)
    left, _ = torch.min(cw + w * (~in_w), dim=-1)
    empty = (right < left) | (bottom < top)
    return torch.stack([left, top, right, bottom], dim=-1) * (~empty).unsqueeze(-1)


def mask_postprocess(low_res, img_size, input_size, original_size, mask_threshold=0.0, stability_offset=1.0):
    """What SamAutomaticMaskGenerator._process_batch does with the decoder's logits (automatic_mask_generator.py
    :281-312, through SamPredictor.predict_torch(return_logits=True) predictor.py:229-238): upscale to the frame,
    stability score on the logits, threshold, boxes.  low_res (B,C,h,w) -> (masks bool (B*C,H,W), stability (B*C,),
    boxes (B*C,4) long)."""
    m = postprocess_masks(low_res, img_size, input_size, original_size).flatten(0, 1)
    st = stability_score(m, mask_threshold, stability_offset)
    mb = m > mask_threshold
    return mb, st, mask_to_box(mb)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms restated from its published algorithm (un-vendored dependency of the reference; PARITY
    UNPINNED at this boundary): visit boxes by decreasing score, drop those whose IoU with a kept one exceeds the
    threshold; area = (x2-x1)*(y2-y1), float32.  Returns kept indices by decreasing score."""
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    b = boxes.float()[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep, dead = [], torch.zeros(len(b), dtype=torch.bool)
    for i in range(len(b)):
        if dead[i]:
            continue
        keep.append(i)
        w = (torch.minimum(b[i, 2], b[:, 2]) - torch.maximum(b[i, 0], b[:, 0])).clamp(min=0)
        h = (torch.minimum(b[i, 3], b[:, 3]) - torch.maximum(b[i, 1], b[:, 1])).clamp(min=0)
        inter = w * h
        dead |= (inter / (area[i] + area - inter)) > iou_threshold
    return order[torch.tensor(keep, dtype=torch.long)]
