"""DINOv2 descriptor model of the ISM stage on MI355X -- drop-in for Instance_Segmentation_Model/model/
{vision_transformer.py, dinov2.py} and the crop pipeline of utils/bbox_utils.py.

Same class names, constructor arguments, state_dict keys and method names as the reference:
``DinoVisionTransformer`` (vision_transformer.py:44-325; ``vit_small/base/large/giant2`` builders :340-392) and
``CustomDINOv2`` (dinov2.py:92-258), so ``model.load_state_dict(torch.load("dinov2_vitl14_pretrain.pth"))``
(dinov2.py:107) and the detector's calls (``descriptor_model.forward(image_np, proposals)``, model/detector.py) are
unchanged.

Hardware-first execution:
  * proposal crops: the reference repeats the normalised frame P times ((P,3,480,640) fp32), multiplies by the masks
    and runs a per-proposal Python loop of slice / interpolate / pad / interpolate.  Here the host mirrors only the
    loop's size arithmetic (a (P,12) int32 record table) and ONE kernel (s6d_crop_resize_pad_f32) reads each
    surviving frame pixel once and writes the (P,3,224,224) crops and the (P,224,224) masks -- bit-exact.
  * ViT-L/14 in bf16 (``S6D_DINO_DTYPE`` = bf16 | fp32): library GEMMs, fused MFMA attention over the 257 tokens
    (s6d_seq_attention_bf16), every residual add folded into the next LayerNorm pass (s6d_add_layernorm_bf16);
    LayerScale is folded into the proj / fc2 weights once per weight version, so it costs nothing at run time.
  * chunking (``chunk_size``) only bounds activation memory; with 288 GB of HBM the default runs a frame's proposals
    in chunks of 128 instead of the reference's 16.
"""
import math
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import policy
from ..utils.linear import eligible, fused_linear, lnfold_cached, lnfold_eligible

descriptor_size = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024, "dinov2_vitg14": 1536}
descriptor_map = {"dinov2_vits14": "vit_small", "dinov2_vitb14": "vit_base", "dinov2_vitl14": "vit_large",
                  "dinov2_vitg14": "vit_giant2"}
RGB_MEAN, RGB_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # dinov2.py:118


def _dtype():
    return {"fp32": torch.float32, "bf16": torch.bfloat16}[policy.current().dino_dtype]


class PatchEmbed(nn.Module):
    """layers/patch_embed.py:26-79 (flatten_embedding=True, no norm)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):                                           # non-overlapping conv == GEMM on unfolded patches
        B, C, H, W = x.shape
        p = self.patch_size[0]
        assert H % p == 0 and W % p == 0, "input size must be a multiple of the patch size"
        x = x.view(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // p) * (W // p), C * p * p)
        return F.linear(x, self.proj.weight.flatten(1).to(x.dtype), self.proj.bias.to(x.dtype))


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma.to(x.dtype)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, N, C))


MemEffAttention = Attention    # the reference falls back to Attention without xformers (layers/attention.py:67-69)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features, bias=bias)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, in_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    """layers/block.py:36-107, inference branch."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, init_values=None,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))

    # -- fused path helpers ---------------------------------------------------------------------------------------
    def _folded(self, dtype):
        """(W_proj, b_proj, b_proj_f32, W_fc2, b_fc2, b_fc2_f32) with the LayerScale gains folded in, cached per weight version."""
        srcs = [self.attn.proj.weight, self.attn.proj.bias, self.mlp.fc2.weight, self.mlp.fc2.bias]
        g1 = self.ls1.gamma if isinstance(self.ls1, LayerScale) else None
        g2 = self.ls2.gamma if isinstance(self.ls2, LayerScale) else None
        key = tuple((t._version, t.data_ptr()) for t in srcs + [g for g in (g1, g2) if g is not None]) + (dtype,)
        c = getattr(self, "_s6d_folded", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                def fold(w, b, g):
                    w, b = w.detach().float(), b.detach().float()
                    if g is not None:
                        w, b = w * g.detach().float()[:, None], b * g.detach().float()
                    return w.to(dtype).contiguous(), b.to(dtype).contiguous(), b.contiguous()
                c = (key,) + fold(srcs[0], srcs[1], g1) + fold(srcs[2], srcs[3], g2)
            self._s6d_folded = c
        return c[1:]


def _ln_f32(norm):
    key = (norm.weight._version, norm.bias._version, norm.weight.data_ptr())
    c = getattr(norm, "_s6d_f32", None)
    if c is None or c[0] != key:
        c = (key, norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous())
        norm._s6d_f32 = c
    return c[1], c[2]


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0, drop_path_uniform=False,
                 init_values=None, embed_layer=PatchEmbed, act_layer=nn.GELU, block_fn=Block, ffn_layer="mlp",
                 block_chunks=1, num_register_tokens=0, interpolate_antialias=False, interpolate_offset=0.1):
        super().__init__()
        if ffn_layer != "mlp" or num_register_tokens != 0 or block_chunks != 0 or drop_path_rate != 0.0:
            raise NotImplementedError("MI355X build covers the released descriptor configuration: mlp FFN, no register "
                                      "tokens, block_chunks=0, inference (dinov2.py:44-58)")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens, self.n_blocks, self.num_heads, self.patch_size = 1, depth, num_heads, patch_size
        self.num_register_tokens, self.register_tokens = 0, None
        self.interpolate_antialias, self.interpolate_offset = interpolate_antialias, interpolate_offset
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.chunked_blocks = False
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values,
                                           norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))

    def interpolate_pos_encoding(self, x, w, h):
        """vision_transformer.py:179-206; the result depends on (w, h) and the weights only -> cached."""
        npatch, N = x.shape[1] - 1, self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed.to(x.dtype)
        key = (w, h, x.dtype, self.pos_embed._version, self.pos_embed.data_ptr())
        c = getattr(self, "_s6d_pos", None)
        if c is None or c[0] != key:
            pos = self.pos_embed.detach().float()
            dim = pos.shape[-1]
            w0, h0 = w // self.patch_size + self.interpolate_offset, h // self.patch_size + self.interpolate_offset
            sq = math.sqrt(N)
            pp = F.interpolate(pos[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2),
                               scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic",
                               antialias=self.interpolate_antialias)
            assert int(w0) == pp.shape[-2] and int(h0) == pp.shape[-1]
            c = (key, torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1).to(x.dtype))
            self._s6d_pos = c
        return c[1]

    def prepare_tokens_with_masks(self, x, masks=None):
        if masks is not None:
            raise NotImplementedError("masked-token training input is outside the inference path")
        B, _, w, h = x.shape
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.to(x.dtype).expand(B, -1, -1), x), dim=1)
        return x + self.interpolate_pos_encoding(x, w, h)

    def _fusable(self, x):
        hd = self.embed_dim // self.num_heads
        return policy.guard("ism.DinoVisionTransformer", cuda=x.is_cuda, bf16=x.dtype == torch.bfloat16, head_dim=hd in (64, 80),
                            have=ops.have("seq_attention") and ops.have("add_layernorm"))

    def _blocks_fused(self, x):
        """Residual stream through all blocks + the final norm; returns (x_prenorm, x_norm)."""
        x = x.contiguous()
        delta = None
        scale = (self.embed_dim // self.num_heads) ** -0.5
        C = x.shape[-1]
        exact_gelu = all(isinstance(blk.mlp.act, nn.GELU) and blk.mlp.act.approximate == "none" for blk in self.blocks)
        rows = x.numel() // C
        if policy.current().dino_gemm in ("fp8", "fp8mx"):
            return self._blocks_fp8(x, scale, exact_gelu)
        if (exact_gelu and C % 256 == 0 and all(lnfold_eligible(x, C, C) and lnfold_eligible(x, blk.mlp.fc1.out_features, C) and
                                                lnfold_eligible(x, C, blk.mlp.fc1.out_features) for blk in self.blocks)
                and rows <= ops.gemm_one_launch_rows(max(3 * C, max(blk.mlp.fc1.out_features for blk in self.blocks)))):
            # neither the residual adds nor the block LayerNorms as passes of their own: the same folded loop as the SAM encoder's
            # (sam/image_encoder.py::_blocks_lnfold), LayerScale already folded into proj / fc2 (Block._folded)
            x = x.clone()
            B, N, _ = x.shape
            x2 = x.view(rows, C)
            blocks = list(self.blocks)
            st = ops.row_stats(x2, blocks[0].norm1.eps)
            sp = torch.empty(C // 32, 2, rows, dtype=torch.float32, device=x.device)
            for i, blk in enumerate(blocks):
                wp, bp, bpf, w2, b2, b2f = blk._folded(x.dtype)
                wf, cs, bf = lnfold_cached(blk.attn.qkv, blk.norm1)
                qkv = ops.gemm_bf16_lnfold(x2, st, wf, cs, bf).view(B, N, 3 * C)
                o = ops.seq_attention(qkv, blk.attn.num_heads, scale)
                ops.gemm_bf16(o.reshape(rows, C), wp, bpf, residual=x2, out=x2, stats_partial=sp)
                st = ops.ln_stats_finalize(sp, 32, blk.norm2.eps)
                w1, c1, b1 = lnfold_cached(blk.mlp.fc1, blk.norm2)
                h = ops.gemm_bf16_lnfold(x2, st, w1, c1, b1, gelu=True)
                last = i + 1 == len(blocks)
                ops.gemm_bf16(h, w2, b2f, residual=x2, out=x2, stats_partial=None if last else sp)
                if not last:
                    st = ops.ln_stats_finalize(sp, 32, blocks[i + 1].norm1.eps)
            g, b = _ln_f32(self.norm)
            return ops.add_layernorm(x, None, g, b, self.norm.eps)
        if ops.have("gemm_bf16_res") and C % 256 == 0 and exact_gelu and all(
                eligible(x, C, C) and eligible(x, C, blk.mlp.fc2.in_features) for blk in self.blocks):
            # round 3: both residual adds of a block in the epilogues of the proj / fc2 GEMMs (in place on the stream tensor),
            # LayerNorms as one-read passes
            x = x.clone()
            M = x.numel() // C
            for blk in self.blocks:
                wp, bp, bpf, w2, b2, b2f = blk._folded(x.dtype)
                g, b = _ln_f32(blk.norm1)
                _, h = ops.add_layernorm(x, None, g, b, blk.norm1.eps)
                o = ops.seq_attention(fused_linear(blk.attn.qkv, h).contiguous(), blk.attn.num_heads, scale)
                ops.gemm_bf16(o, wp, bpf, residual=x, out=x.view(M, C))
                g, b = _ln_f32(blk.norm2)
                _, h = ops.add_layernorm(x, None, g, b, blk.norm2.eps)
                ops.gemm_bf16(fused_linear(blk.mlp.fc1, h, gelu=True), w2, b2f, residual=x, out=x.view(M, C))
            g, b = _ln_f32(self.norm)
            return ops.add_layernorm(x, None, g, b, self.norm.eps)
        for blk in self.blocks:
            wp, bp, bpf, w2, b2, b2f = blk._folded(x.dtype)
            own = eligible(x, wp.shape[0], wp.shape[1]) and eligible(x, w2.shape[0], w2.shape[1])   # s6d_gemm_bf16
            g, b = _ln_f32(blk.norm1)
            x, h = ops.add_layernorm(x, delta, g, b, blk.norm1.eps)
            o = ops.seq_attention(fused_linear(blk.attn.qkv, h).contiguous(), blk.attn.num_heads, scale)
            a = ops.gemm_bf16(o, wp, bpf) if own else F.linear(o, wp, bp)
            g, b = _ln_f32(blk.norm2)
            x, h = ops.add_layernorm(x, a, g, b, blk.norm2.eps)
            if own and isinstance(blk.mlp.act, nn.GELU) and blk.mlp.act.approximate == "none":
                delta = ops.gemm_bf16(fused_linear(blk.mlp.fc1, h, gelu=True), w2, b2f)
            else:
                delta = F.linear(blk.mlp.act(blk.mlp.fc1(h)), w2, b2)
        g, b = _ln_f32(self.norm)
        return ops.add_layernorm(x, delta, g, b, self.norm.eps)

    def _blocks_fp8(self, x, scale, exact_gelu):
        """BASELINE configs[4] for the descriptor ViT (round 4; opt-in: S6D_DINO_GEMM=fp8, never the default): the two LayerNorm-fed
        GEMMs of every block -- qkv and fc1 + GELU, 58 % of the block's GEMM FLOP -- on the fp8 matrix cores, exactly as the SAM
        encoder's fp8 loop (sam/image_encoder.py::_blocks_fp8): LayerNorm writes e4m3 rows with one power-of-two scale per token
        (s6d_layernorm_fp8), the weights carry one per output channel (utils/fp8.py), both ride in the matrix instruction's block
        scales (s6d_gemm_fp8).  proj and fc2 (LayerScale folded in) stay bf16 with the residual add in their epilogue."""
        from ..utils import fp8
        C = x.shape[-1]
        if not (exact_gelu and ops.have("gemm_fp8") and ops.have("layernorm_fp8") and ops.have("gemm_bf16_res") and x.is_cuda
                and x.dtype == torch.bfloat16 and C % 256 == 0 and C <= 2048):
            raise RuntimeError("S6D_DINO_GEMM=fp8 needs the fp8 kernels of libsam6d_hip.so, a bf16 device token stream, exact GELU "
                               "and 256 | C <= 2048")
        x = x.clone()
        B, N, _ = x.shape
        rows = B * N
        x2 = x.view(rows, C)
        # S6D_DINO_GEMM=fp8mx: fc2 too, fed by fc1's MX-scaled e4m3 output (s6d_gemm_fp8_gelu_mx -> s6d_gemm_fp8_mxa, as the SAM
        # encoder's fp8mx loop); fc2's bf16 output is the delta the next quantising LayerNorm adds
        mx = policy.current().dino_gemm == "fp8mx" and ops.have("gemm_fp8_mx")
        delta = None
        for blk in self.blocks:
            wp, bp, bpf, w2, b2, b2f = blk._folded(x.dtype)
            g, b = _ln_f32(blk.norm1)
            if delta is None:
                h8, hs = ops.layernorm_fp8(x2, g, b, blk.norm1.eps)
            else:
                x2, h8, hs = ops.layernorm_fp8(x2, g, b, blk.norm1.eps, delta=delta)
            wq, ws, bq = fp8.cached_weight(blk.attn.qkv)
            qkv = ops.gemm_fp8(h8, hs, wq, ws, bq).view(B, N, 3 * C)
            o = ops.seq_attention(qkv, blk.attn.num_heads, scale)
            ops.gemm_bf16(o.reshape(rows, C), wp, bpf, residual=x2, out=x2)
            g, b = _ln_f32(blk.norm2)
            h8, hs = ops.layernorm_fp8(x2, g, b, blk.norm2.eps)
            w1, s1, b1 = fp8.cached_weight(blk.mlp.fc1)
            if mx:
                q8, qs = ops.gemm_fp8_gelu_mx(h8, hs, w1, s1, b1)
                w2q, w2s = self._fc2_fp8(blk, w2)
                delta = ops.gemm_fp8_mxa(q8, qs, w2q, w2s, b2f)
            else:
                ops.gemm_bf16(ops.gemm_fp8(h8, hs, w1, s1, b1, gelu=True), w2, b2f, residual=x2, out=x2)
        g, b = _ln_f32(self.norm)
        return ops.add_layernorm(x2.view(B, N, C), None if delta is None else delta.view(B, N, C), g, b, self.norm.eps)

    @staticmethod
    def _fc2_fp8(blk, w2_folded):
        """e4m3 bytes + per-channel scales of fc2 with the LayerScale gain folded in (Block._folded), cached on the folded tensor."""
        from ..utils import fp8
        c = getattr(blk, "_s6d_fc2_fp8", None)
        key = (w2_folded.data_ptr(), w2_folded._version)
        if c is None or c[0] != key:
            c = (key,) + fp8.quantize_rows(w2_folded.float())
            blk._s6d_fc2_fp8 = c
        return c[1], c[2]

    def forward_features(self, x, masks=None):
        dt = _dtype() if x.is_cuda else torch.float32
        x = x.to(dt)
        with torch.autocast(device_type=x.device.type, dtype=dt, enabled=dt != torch.float32):
            x = self.prepare_tokens_with_masks(x, masks)
            if self._fusable(x):
                x, xn = self._blocks_fused(x)
            else:
                for blk in self.blocks:
                    x = blk(x)
                xn = self.norm(x)
        return {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:1], "x_norm_patchtokens": xn[:, 1:],
                "x_prenorm": x, "masks": masks}

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret["x_norm_clstoken"])


def _vit(embed_dim, depth, num_heads, patch_size=16, num_register_tokens=0, **kwargs):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                                 mlp_ratio=4, num_register_tokens=num_register_tokens, **kwargs)


vit_small = partial(_vit, 384, 12, 6)
vit_base = partial(_vit, 768, 12, 12)
vit_large = partial(_vit, 1024, 24, 16)
vit_giant2 = partial(_vit, 1536, 40, 24)
_ARCH = {"vit_small": vit_small, "vit_base": vit_base, "vit_large": vit_large, "vit_giant2": vit_giant2}


def _make_dinov2_model(*, arch_name="vit_large", img_size=518, patch_size=14, init_values=1.0, ffn_layer="mlp",
                       block_chunks=0, num_register_tokens=0, interpolate_antialias=False, interpolate_offset=0.1,
                       pretrained=False, **kwargs):
    """dinov2.py:44-88 without the download branch (no network on the serving hosts)."""
    if pretrained:
        raise NotImplementedError("load the checkpoint from disk: CustomDINOv2(checkpoint_dir=...)")
    return _ARCH[arch_name](patch_size=patch_size, img_size=img_size, init_values=init_values, ffn_layer=ffn_layer,
                            block_chunks=block_chunks, num_register_tokens=num_register_tokens,
                            interpolate_antialias=interpolate_antialias, interpolate_offset=interpolate_offset, **kwargs)


class _FrameBatcher:
    """The crops of a group's frames go through the ViT in batches chosen for the GEMM tile grid (plan_chunks), not frame by frame:
    128 proposals x 257 tokens are 128.5 row tiles, so a frame's proj / fc2 GEMMs run three rounds of tiles for two rounds of work
    (measured: 25.6 ms per 128 crops alone, 21.9 ms in batches of 255; inside a frame group the gain shrinks to 0.7 ms per frame, the
    full batches running at the socket's power limit).  A full batch is launched as soon as 255 crops have accumulated -- the device
    then has descriptor work queued while the host prepares the next frame's proposals -- the rest at ``finish``.  A row of
    a batch does not depend on its neighbours in any kernel of the path: the values are those of per-frame calls."""

    FULL = 255                                                            # 255 x 257 token rows = 256 row tiles exactly

    def __init__(self, desc):
        self.desc, self.counts, self.rgbs, self.masks, self.outs, self.pending = desc, [], [], [], [], 0

    def _run(self, c):
        rgbs = self.rgbs[0] if len(self.rgbs) == 1 else torch.cat(self.rgbs)
        masks = self.masks[0] if len(self.masks) == 1 else torch.cat(self.masks)
        self.outs.append(self.desc.compute_cls_and_patch_features(rgbs[:c], masks[:c]))
        self.rgbs, self.masks = ([rgbs[c:]], [masks[c:]]) if c < rgbs.shape[0] else ([], [])
        self.pending -= c

    @torch.no_grad()
    def add(self, image_np, proposals):
        n = int(proposals.masks.shape[0])
        self.counts.append(n)
        if n == 0:
            return
        r, m = self.desc._crops(image_np, proposals.masks, proposals.boxes, True, True)
        self.rgbs.append(r)
        self.masks.append(m)
        self.pending += n
        while self.pending >= self.FULL:
            self._run(self.FULL)

    @torch.no_grad()
    def finish(self):
        for c in plan_chunks(self.pending):
            self._run(c)
        if not self.outs:
            return [None] * len(self.counts)
        cls = self.outs[0][0] if len(self.outs) == 1 else torch.cat([o[0] for o in self.outs])
        patch = self.outs[0][1] if len(self.outs) == 1 else torch.cat([o[1] for o in self.outs])
        res, at = [], 0
        for k in self.counts:
            res.append((cls[at:at + k], patch[at:at + k]) if k else None)
            at += k
        return res


_PLAN = {}


def plan_chunks(n, tokens=257, dim=1024, hidden=4096, cus=256, max_chunk=255):
    """Split n proposals into ViT batches that fill the GEMM tile grid of the MI355X: a batch of c proposals is ceil(c * tokens / 256)
    row tiles; qkv / proj / fc1 / fc2 run (3 dim, dim, hidden, dim) / 256 column tiles each, one 256 x 256 tile per CU and round, a
    round costing ~ K / 64 K tiles (+ epilogue).  Dynamic programme over the first batch's size, minimising the rounds' cost plus a
    small per-batch launch overhead; 255 x 257 rows = 256 row tiles exactly is the sweet spot.  The table grows on demand and is
    shared between calls.  -> list of batch sizes (largest first), sum n."""
    key = (tokens, dim, hidden, cus, max_chunk)
    tab = _PLAN.get(key)
    if tab is None:
        def cost(c):
            mt = -(-c * tokens // 256)
            rounds = lambda nt: -(-mt * nt // cus)                         # noqa: E731
            kt = lambda k: k / 64.0 + 4.0                                  # noqa: E731  K tiles + an epilogue's worth
            return (rounds(3 * dim // 256) * kt(dim) + rounds(dim // 256) * kt(dim) + rounds(hidden // 256) * kt(dim)
                    + rounds(dim // 256) * kt(hidden)) + 6.0               # + launches of a batch
        tab = _PLAN[key] = dict(cost=[0.0] + [cost(c) for c in range(1, max_chunk + 1)], best=[0.0], first=[0])
    cost, best, first = tab["cost"], tab["best"], tab["first"]
    for m in range(len(best), n + 1):
        bv, bc = float("inf"), 0
        for c in range(1, min(max_chunk, m) + 1):
            v = cost[c] + best[m - c]
            if v < bv - 1e-9:
                bv, bc = v, c
        best.append(bv)
        first.append(bc)
    out, m = [], n
    while m > 0:
        out.append(first[m])
        m -= first[m]
    out.sort(reverse=True)
    return out


def _crop_geometry(boxes, target):
    """Size arithmetic of CropResizePad.__call__ (utils/bbox_utils.py:98-126) for every proposal, vectorised, plus the
    three ways the reference fails on a box: empty box / a side that vanishes after the first resize (F.interpolate
    raises), a non-square padded crop (its assert), a second resize that does not reach `target` (torch.stack raises).
    Mirrors the reference's arithmetic type by type: scale = float32(1 / longest side) * float32(target) read back as a
    Python float; sizes floor(size * scale) in double; the index scale handed to the resize is float32(1 / scale)."""
    b = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    empty = np.minimum(w, h) <= 0
    ws, hs = np.maximum(w, 1), np.maximum(h, 1)
    # `self.target_max / torch.max(box_sizes)`: int / LongTensor is Tensor.__rtruediv__ = reciprocal() * other, i.e.
    # float32(1 / side) * float32(target) -- one ulp away from float32(target / side) for some sides (446 -> 224 rows,
    # not 223), so it is spelled the reference's way
    s1 = ((np.float32(1.0) / np.maximum(ws, hs).astype(np.float32)) * np.float32(target)).astype(np.float64)
    h1, w1 = np.floor(hs * s1).astype(np.int64), np.floor(ws * s1).astype(np.int64)
    vanish = np.minimum(h1, w1) <= 0
    h1s, w1s = np.maximum(h1, 1), np.maximum(w1, 1)
    padded = (w1s / h1s) != 1.0                                      # `self.target_ratio != original_ratio`
    top = np.where(padded, np.maximum((target - h1s) // 2, 0), 0)
    left = np.where(padded, np.maximum((target - w1s) // 2, 0), 0)
    Hp = np.where(padded, h1s + top + (target - h1s - top), h1s)      # F.pad with the reference's bottom / right amounts
    Wp = np.where(padded, w1s + left + (target - w1s - left), w1s)
    notsquare = Hp != Wp
    s2 = target / np.maximum(Hp, 1).astype(np.float64)
    short = np.floor(Hp * s2).astype(np.int64) != target
    return dict(b=b, h=h, w=w, h1=h1s, w1=w1s, top=top, left=left, Hp=Hp, s1=s1, s2=s2, empty=empty, vanish=vanish,
                notsquare=notsquare, short=short)


def crop_valid(boxes, target):
    """(P,) bool: the proposals whose crop the reference can produce (see crop_params for the three failure modes)."""
    g = _crop_geometry(boxes, target)
    return ~(g["empty"] | g["vanish"] | g["notsquare"] | g["short"])


def crop_params(boxes, target):
    """Geometry of CropResizePad.__call__ for every proposal, as the (P,12) int32 record table of
    s6d_crop_resize_pad_f32.  boxes (P,4) integer xyxy on the HOST.  Raises where the reference does: RuntimeError for an
    empty box or a side that vanishes in the first resize, AssertionError for a non-square padded crop, RuntimeError
    ("stack expects each tensor to be equal size") for a crop whose second resize does not land on `target`."""
    g = _crop_geometry(boxes, target)
    P = g["b"].shape[0]
    rec = np.zeros((P, 12), dtype=np.int32)
    if P == 0:
        return rec
    if g["empty"].any():
        raise RuntimeError("crop_params: empty proposal box (the reference fails in F.interpolate on these)")
    if g["vanish"].any():
        raise RuntimeError("crop_params: a proposal side vanishes after the resize (the reference fails here too)")
    if g["notsquare"].any():
        raise AssertionError("image is not square after padding")   # bbox_utils.py:120-122
    if g["short"].any():
        raise RuntimeError("stack expects each tensor to be equal size: a crop's second resize does not reach "
                           f"{target} (reference behaviour for this box shape, bbox_utils.py:123-126)")
    b = g["b"]
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = b[:, 0], b[:, 1], g["h"], g["w"]
    rec[:, 4], rec[:, 5], rec[:, 6], rec[:, 7], rec[:, 8] = g["h1"], g["w1"], g["top"], g["left"], g["Hp"]
    rec[:, 9] = (1.0 / g["s1"]).astype(np.float32).view(np.int32)
    rec[:, 10] = (1.0 / g["s2"]).astype(np.float32).view(np.int32)
    return rec


class CustomDINOv2(nn.Module):
    """dinov2.py:92-258 (pl.LightningModule there; nothing Lightning-specific is used at inference)."""

    def __init__(self, model_name, token_name, image_size, chunk_size, descriptor_width_size, checkpoint_dir,
                 patch_size=14, validpatch_thresh=0.5):
        super().__init__()
        self.model_name = model_name
        self.model = _make_dinov2_model(arch_name=descriptor_map[model_name], pretrained=False)
        if checkpoint_dir is not None:                               # None: weights are loaded by the caller
            self.model.load_state_dict(torch.load(os.path.join(checkpoint_dir, f"{model_name}_pretrain.pth")))
        self.validpatch_thresh, self.token_name = validpatch_thresh, token_name
        self.chunk_size, self.patch_size, self.proposal_size = chunk_size, patch_size, image_size
        self.descriptor_width_size = descriptor_width_size

    # ---- proposal crops -----------------------------------------------------------------------------------------
    def _crops(self, image_np, masks, boxes, rgb, mask):
        if not ops.have("crop_resize_pad"):
            raise RuntimeError("libsam6d_hip.so lacks s6d_crop_resize_pad_f32")
        m = masks.float()
        if m.dim() == 4:
            m = m.squeeze(1)
        params = torch.from_numpy(crop_params(boxes.detach().cpu().numpy(), self.proposal_size)).to(m.device)
        # image_np: the reference's numpy frame (H,W,3) uint8 -- or the same frame as a device tensor (FramePipeline hands it over
        # without the device -> host -> device round trip)
        if not rgb:
            img = None
        elif torch.is_tensor(image_np):
            img = image_np.to(device=m.device, dtype=torch.uint8).contiguous()
        else:
            img = torch.as_tensor(np.ascontiguousarray(image_np)).to(m.device)
        return ops.crop_resize_pad(img, m.contiguous(), params, self.proposal_size, RGB_MEAN, RGB_STD, rgb=rgb, mask=mask)

    def process_rgb_proposals(self, image_np, masks, boxes):
        return self._crops(image_np, masks, boxes, True, False)[0]

    def process_masks_proposals(self, masks, boxes):
        return self._crops(None, masks, boxes, False, True)[1]

    # ---- descriptors --------------------------------------------------------------------------------------------
    def _chunks(self, n):
        return [(i, min(i + self.chunk_size, n)) for i in range(0, n, self.chunk_size)]

    @torch.no_grad()
    def compute_features(self, images, token_name):
        if token_name != "x_norm_clstoken":
            raise NotImplementedError
        return torch.cat([self.model(images[a:b]).float() for a, b in self._chunks(images.shape[0])])

    def forward_by_chunk(self, processed_rgbs):
        return self.compute_features(processed_rgbs, "x_norm_clstoken")

    @torch.no_grad()
    def forward_cls_token(self, image_np, proposals):
        return self.forward_by_chunk(self.process_rgb_proposals(image_np, proposals.masks, proposals.boxes))

    @torch.no_grad()
    def compute_cls_and_patch_features(self, images, masks):
        f = self.model(images, is_training=True)
        keep = F.avg_pool2d(masks, self.patch_size, self.patch_size).flatten(-2) > self.validpatch_thresh
        patch = F.normalize(f["x_norm_patchtokens"].float() * keep.unsqueeze(-1), dim=-1)
        return f["x_norm_clstoken"].float(), patch

    @torch.no_grad()
    def compute_masked_patch_feature(self, images, masks):
        return torch.cat([self.compute_cls_and_patch_features(images[a:b], masks[a:b])[1]
                          for a, b in self._chunks(images.shape[0])])

    def forward_by_chunk_v2(self, processed_rgbs, masks):
        return self.compute_masked_patch_feature(processed_rgbs, masks)

    @torch.no_grad()
    def forward_patch_tokens(self, image_np, proposals):
        rgbs, masks = self._crops(image_np, proposals.masks, proposals.boxes, True, True)
        return self.compute_masked_patch_feature(rgbs, masks)

    def frame_batcher(self):
        """Descriptors for the frames of a launch group with the ViT batched ACROSS the frames: ``add(image, proposals)`` per frame,
        ``finish()`` -> list of (cls, patch) per frame (see _FrameBatcher)."""
        return _FrameBatcher(self)

    @torch.no_grad()
    def forward_frames(self, images, proposals_list):
        """``forward`` for the frames of a launch group in one go -> list of (cls, patch) per frame, the values of per-frame calls."""
        fb = self.frame_batcher()
        for img, pr in zip(images, proposals_list):
            fb.add(img, pr)
        return fb.finish()

    @torch.no_grad()
    def forward(self, image_np, proposals):
        """(cls (P,C), masked + normalised patch descriptors (P,256,C)), one crop kernel + chunked ViT passes."""
        rgbs, masks = self._crops(image_np, proposals.masks, proposals.boxes, True, True)
        outs = [self.compute_cls_and_patch_features(rgbs[a:b], masks[a:b]) for a, b in self._chunks(rgbs.shape[0])]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
