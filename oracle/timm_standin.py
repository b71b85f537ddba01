"""Stand-in for ``timm.models.vision_transformer.VisionTransformer`` (test infra).

The reference PEM feature extractor subclasses timm's ViT
(reference: Pose_Estimation_Model/model/feature_extraction.py:7,17-35,51-57) but
timm is neither vendored nor pinned (SAM-6D/environment.yaml:34 ``- timm``,
Pose_Estimation_Model/dependencies.sh:4) and is absent from this image.  This
module restates timm's published ``VisionTransformer`` defaults -- the only
members the reference touches are ``patch_embed``, ``_pos_embed``, ``norm_pre``,
``blocks`` and ``norm`` -- so that the reference file can be imported for golden
generation.  PARITY UNPINNED at this boundary: no reference test fixes timm's
numerics; attention is written as explicit softmax(q k^T * hd^-0.5) v.

Layout restated (timm >= 0.6 naming, which the released checkpoint follows):
  cls_token (1,1,D), pos_embed (1,1+N,D) added after the cls concat,
  patch_embed.proj = Conv2d(3,D,16,16), no pre-norm, pre-LN blocks
  {norm1, attn.qkv (fused, bias), attn.proj, norm2, mlp.fc1, GELU(exact), mlp.fc2},
  final norm (eps from norm_layer), head = Linear(D, 1000) (unused on this path).
"""
import sys
import types

import torch
import torch.nn as nn


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, norm_layer=None, **_):
        super().__init__()
        norm_layer = norm_layer or (lambda d: nn.LayerNorm(d, eps=1e-6))
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * 0.02)
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[
            _Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)

    def _pos_embed(self, x):
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        return x + self.pos_embed


def install():
    """Register ``timm.models.vision_transformer`` in sys.modules (if timm is absent)."""
    if "timm" in sys.modules:
        return
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.VisionTransformer = VisionTransformer
    models.vision_transformer = vt
    timm.models = models
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt})
