"""CPU restatement of the DINOv2 descriptor path of the ISM stage -- TEST INFRASTRUCTURE ONLY.

Functional torch-fp32 restatement of
  * ``Instance_Segmentation_Model/model/vision_transformer.py`` DinoVisionTransformer
    (interpolate_pos_encoding :179-206, prepare_tokens_with_masks :208-227, forward_features :250-266),
    ``model/layers/{attention.py:49-62, block.py:82-107, mlp.py, layer_scale.py:23-28, patch_embed.py:66-79}``
  * ``model/dinov2.py`` CustomDINOv2: process_rgb_proposals :131-144, process_masks_proposals :178-189,
    compute_cls_and_patch_features :249-258, compute_masked_patch_feature :215-225
  * ``utils/bbox_utils.py`` CropResizePad.__call__ :98-126
  * torchvision's ``ToTensor`` + ``Normalize`` (un-vendored third party, any version: uint8 HWC -> float CHW / 255,
    then (x - mean) / std), which CustomDINOv2.rgb_normalize composes (dinov2.py:115-120).
Weights: flat {state_dict key: tensor} with the reference key names.  Pinned by tests/golden/dinov2.npz
(reference modules imported unmodified by oracle/gen_golden.py).
"""
import math

import torch
import torch.nn.functional as F

# _make_dinov2_model(arch_name="vit_large") defaults (dinov2.py:44-58): img 518, patch 14, LayerScale, no registers
VIT_L14 = dict(img_size=518, patch=14, dim=1024, depth=24, heads=16, offset=0.1)
MINI = dict(img_size=70, patch=14, dim=128, depth=2, heads=2, offset=0.1)       # head dim 64 like ViT-L
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def rgb_normalize(image_u8):
    """ToTensor + Normalize on an (H,W,3) uint8 array -> (3,H,W) float32."""
    x = torch.as_tensor(image_u8).permute(2, 0, 1).float().div(255)
    m = torch.tensor(MEAN).view(3, 1, 1)
    s = torch.tensor(STD).view(3, 1, 1)
    return x.sub(m).div(s)


def crop_resize_pad(images, boxes, target=224):
    """CropResizePad.__call__ for a square target: per proposal crop box [x1,y1,x2,y2), nearest-resize by
    target / longest side, zero-pad to the square, nearest-resize to target."""
    box_sizes = boxes[:, 2:] - boxes[:, :2]
    scale_factor = target / torch.max(box_sizes, dim=-1)[0]
    out = []
    for image, box, scale in zip(images, boxes, scale_factor):
        image = image[:, box[1]:box[3], box[0]:box[2]]
        image = F.interpolate(image.unsqueeze(0), scale_factor=scale.item())[0]
        h, w = image.shape[1:]
        if 1.0 != w / h:
            top = max((target - h) // 2, 0)
            left = max((target - w) // 2, 0)
            image = F.pad(image, (left, target - w - left, top, target - h - top))
        assert image.shape[1] == image.shape[2]
        image = F.interpolate(image.unsqueeze(0), scale_factor=target / image.shape[1])[0]
        out.append(image)
    return torch.stack(out)


def process_rgb_proposals(image_u8, masks, boxes, target=224):
    """(H,W,3) uint8, masks (P,H,W) float {0,1}, boxes (P,4) long -> (P,3,target,target)."""
    rgb = rgb_normalize(image_u8)
    return crop_resize_pad(rgb.unsqueeze(0) * masks.unsqueeze(1), boxes, target)


def process_masks_proposals(masks, boxes, target=224):
    return crop_resize_pad(masks.unsqueeze(1), boxes, target).squeeze(1)


def interpolate_pos_encoding(W, npatch, w, h, cfg):
    pos = W["pos_embed"].float()
    N = pos.shape[1] - 1
    if npatch == N and w == h:
        return pos
    dim = pos.shape[-1]
    w0, h0 = w // cfg["patch"] + cfg["offset"], h // cfg["patch"] + cfg["offset"]
    sq = math.sqrt(N)
    pp = F.interpolate(pos[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2),
                       scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic", antialias=False)
    assert int(w0) == pp.shape[-2] and int(h0) == pp.shape[-1]
    return torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def attention(W, p, x, heads):
    """Attention.forward (layers/attention.py:49-62): q is scaled BEFORE the product."""
    B, N, C = x.shape
    qkv = F.linear(x, W[p + ".qkv.weight"], W[p + ".qkv.bias"]).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    return F.linear((a @ v).transpose(1, 2).reshape(B, N, C), W[p + ".proj.weight"], W[p + ".proj.bias"])


def block(W, p, x, heads):
    """Block.forward, eval branch (layers/block.py:103-105) with LayerScale (layer_scale.py:27)."""
    h = F.layer_norm(x, (x.shape[-1],), W[p + ".norm1.weight"], W[p + ".norm1.bias"], 1e-6)
    x = x + attention(W, p + ".attn", h, heads) * W[p + ".ls1.gamma"]
    h = F.layer_norm(x, (x.shape[-1],), W[p + ".norm2.weight"], W[p + ".norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, W[p + ".mlp.fc1.weight"], W[p + ".mlp.fc1.bias"])),
                 W[p + ".mlp.fc2.weight"], W[p + ".mlp.fc2.bias"])
    return x + h * W[p + ".ls2.gamma"]


def forward_features(W, x, cfg=VIT_L14):
    """DinoVisionTransformer.forward_features (masks=None): dict with x_norm_clstoken (B,C),
    x_norm_patchtokens (B,N,C), x_prenorm (B,1+N,C)."""
    B, _, w, h = x.shape
    t = F.conv2d(x, W["patch_embed.proj.weight"], W["patch_embed.proj.bias"], stride=cfg["patch"]).flatten(2).transpose(1, 2)
    t = torch.cat((W["cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + interpolate_pos_encoding(W, t.shape[1] - 1, w, h, cfg)
    for i in range(cfg["depth"]):
        t = block(W, f"blocks.{i}", t, cfg["heads"])
    n = F.layer_norm(t, (t.shape[-1],), W["norm.weight"], W["norm.bias"], 1e-6)
    return {"x_norm_clstoken": n[:, 0], "x_norm_patchtokens": n[:, 1:], "x_prenorm": t}


def cls_and_patch_features(W, images, masks, cfg=VIT_L14, thresh=0.5):
    """compute_cls_and_patch_features: cls token; patch tokens zeroed where the 14x14 mean of the proposal mask
    is <= thresh, then L2-normalised (zero rows stay zero: F.normalize eps)."""
    f = forward_features(W, images, cfg)
    keep = F.avg_pool2d(masks, cfg["patch"], cfg["patch"]).flatten(-2) > thresh
    patch = F.normalize(f["x_norm_patchtokens"] * keep.unsqueeze(-1), dim=-1)
    return f["x_norm_clstoken"], patch
