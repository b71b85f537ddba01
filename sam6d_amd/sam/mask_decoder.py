"""SAM prompt encoder + two-way mask decoder on MI355X -- drop-in for segment_anything/modeling/
{prompt_encoder.py, transformer.py, mask_decoder.py} (first slice: the 3.7 TFLOP/frame term).

Same class names, constructor arguments, forward signatures and state_dict keys as the reference
(``PromptEncoder`` prompt_encoder.py:16-166, ``PositionEmbeddingRandom`` :169-214, ``TwoWayTransformer`` /
``TwoWayAttentionBlock`` / ``Attention`` transformer.py:16-240, ``MaskDecoder`` / ``MLP`` mask_decoder.py:16-176), so
``build_sam._build_sam`` (build_sam.py:54-107) can construct them unchanged and ``sam.load_state_dict`` loads the released
checkpoint strictly; ``SamPredictor.predict_torch`` (predictor.py:169-243) and ``Sam.forward`` call them as before.

Two execution paths:
  * library path (any shape, fp32 or autocast): the reference's op sequence on channels-last token tensors.
  * restructured path (``MaskDecoder._predict_masks_shared``; the automatic mask generator's case: point / box prompts,
    no mask inputs, so the dense embedding is one broadcast vector): exact algebra, less work --
      - the image tokens entering layer 0 are the SAME for every prompt: their k / v / q projections (and the
        positional-encoding parts of every later projection, W(x + pe) = Wx + W pe) are computed once per image, not
        once per prompt;
      - projections that read the same tensor are one GEMM (layer 1: [k | v | q_image->token]; final: [k | v | first
        transposed conv], a 2x2/2 transposed conv being a GEMM with 4x the output channels);
      - image->token attention has only n_tok <= 8 keys per head: out_proj(softmax(q k^T) v) = softmax(q k^T) (v W_o^T),
        i.e. one (4096 x 64) x (64 x 256) product per prompt with W_o folded into the 7 value rows.
    As written the decoder costs ~3.6 GFLOP per prompt; restructured ~2.2 GFLOP.
Compute dtype: ``S6D_SAM_DECODER_DTYPE`` = bf16 (default, autocast: GEMMs bf16, LayerNorm / softmax fp32) | fp32.
"""
import math
import os
from typing import Optional, Tuple, Type

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import policy
from ..utils.linear import CastCachedLinear
from .image_encoder import LayerNorm2d, MLPBlock


def _dtype():
    return {"fp32": torch.float32, "bf16": torch.bfloat16}[policy.current().sam_decoder_dtype]


def build_sam_decoder(prompt_embed_dim=256, image_size=1024, vit_patch_size=16):
    """prompt encoder + mask decoder exactly as build_sam._build_sam constructs them (build_sam.py:64-101): a module with
    ``.prompt_encoder`` and ``.mask_decoder`` whose state_dict keys are the ``Sam`` checkpoint's."""
    e = image_size // vit_patch_size
    m = nn.Module()
    m.prompt_encoder = PromptEncoder(embed_dim=prompt_embed_dim, image_embedding_size=(e, e),
                                     input_image_size=(image_size, image_size), mask_in_chans=16)
    m.mask_decoder = MaskDecoder(num_multimask_outputs=3,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=prompt_embed_dim, mlp_dim=2048, num_heads=8),
                                 transformer_dim=prompt_embed_dim, iou_head_depth=3, iou_head_hidden_dim=256)
    return m.eval()


# ---- prompt encoder ------------------------------------------------------------------------------------------------
class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None) -> None:
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))

    def _pe_encoding(self, coords):
        coords = (2 * coords - 1) @ self.positional_encoding_gaussian_matrix
        coords = 2 * np.pi * coords
        return torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)

    def forward(self, size: Tuple[int, int]):
        h, w = size
        grid = torch.ones((h, w), device=self.positional_encoding_gaussian_matrix.device, dtype=torch.float32)
        y = (grid.cumsum(dim=0) - 0.5) / h
        x = (grid.cumsum(dim=1) - 0.5) / w
        return self._pe_encoding(torch.stack([x, y], dim=-1)).permute(2, 0, 1)

    def forward_with_coords(self, coords_input, image_size: Tuple[int, int]):
        coords = coords_input.clone()
        coords[:, :, 0] = coords[:, :, 0] / image_size[1]
        coords[:, :, 1] = coords[:, :, 1] / image_size[0]
        return self._pe_encoding(coords.to(torch.float))


class PromptEncoder(nn.Module):
    def __init__(self, embed_dim: int, image_embedding_size: Tuple[int, int], input_image_size: Tuple[int, int],
                 mask_in_chans: int, activation: Type[nn.Module] = nn.GELU) -> None:
        super().__init__()
        self.embed_dim, self.input_image_size, self.image_embedding_size = embed_dim, input_image_size, image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(4)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans // 4), activation(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans), activation(),
            nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1))
        self.no_mask_embed = nn.Embedding(1, embed_dim)

    def get_dense_pe(self):
        """(1,C,h,w); depends on the buffer only -> cached per buffer version."""
        G = self.pe_layer.positional_encoding_gaussian_matrix
        key = (G._version, G.data_ptr(), G.device)
        c = getattr(self, "_s6d_dense_pe", None)
        if c is None or c[0] != key:
            c = (key, self.pe_layer(self.image_embedding_size).unsqueeze(0))
            self._s6d_dense_pe = c
        return c[1]

    def _embed_points(self, points, labels, pad: bool):
        points = points + 0.5
        if pad:
            points = torch.cat([points, torch.zeros((points.shape[0], 1, 2), device=points.device)], dim=1)
            labels = torch.cat([labels, -torch.ones((labels.shape[0], 1), device=labels.device)], dim=1)
        e = self.pe_layer.forward_with_coords(points, self.input_image_size)
        # masked writes without host syncs (the reference's boolean-index assignments, prompt_encoder.py:86-89)
        lab = labels.unsqueeze(-1)
        e = torch.where(lab == -1, self.not_a_point_embed.weight.expand_as(e), e)
        e = e + (lab == 0) * self.point_embeddings[0].weight + (lab == 1) * self.point_embeddings[1].weight
        return e

    def _embed_boxes(self, boxes):
        coords = (boxes + 0.5).reshape(-1, 2, 2)
        e = self.pe_layer.forward_with_coords(coords, self.input_image_size)
        return e + torch.stack([self.point_embeddings[2].weight, self.point_embeddings[3].weight], dim=1)

    def _get_device(self):
        return self.point_embeddings[0].weight.device

    def forward(self, points, boxes, masks):
        if points is not None:
            bs = points[0].shape[0]
        elif boxes is not None:
            bs = boxes.shape[0]
        elif masks is not None:
            bs = masks.shape[0]
        else:
            bs = 1
        sparse = torch.empty((bs, 0, self.embed_dim), device=self._get_device())
        if points is not None:
            sparse = torch.cat([sparse, self._embed_points(points[0], points[1], pad=(boxes is None))], dim=1)
        if boxes is not None:
            sparse = torch.cat([sparse, self._embed_boxes(boxes)], dim=1)
        if masks is not None:
            dense = self.mask_downscaling(masks)
        else:
            dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(bs, -1, self.image_embedding_size[0],
                                                                          self.image_embedding_size[1])
        return sparse, dense


# ---- two-way transformer ---------------------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1) -> None:
        super().__init__()
        self.embedding_dim, self.internal_dim, self.num_heads = embedding_dim, embedding_dim // downsample_rate, num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        # CastCachedLinear = nn.Linear with its autocast weight casts kept across calls (checkpoint keys unchanged)
        self.q_proj = CastCachedLinear(embedding_dim, self.internal_dim)
        self.k_proj = CastCachedLinear(embedding_dim, self.internal_dim)
        self.v_proj = CastCachedLinear(embedding_dim, self.internal_dim)
        self.out_proj = CastCachedLinear(self.internal_dim, embedding_dim)

    def _heads(self, x):
        b, n, c = x.shape
        return x.reshape(b, n, self.num_heads, c // self.num_heads).transpose(1, 2)

    def attend(self, q, k, v):
        """Projected q (B,Nq,Ci), k, v (B|1,Nk,Ci) -> out_proj(softmax(q k^T / sqrt(d)) v)."""
        q, k, v = self._heads(q), self._heads(k), self._heads(v)
        a = torch.softmax((q @ k.transpose(-1, -2)).float() / math.sqrt(q.shape[-1]), dim=-1).to(v.dtype)
        o = (a @ v).transpose(1, 2)
        return self.out_proj(o.reshape(o.shape[0], o.shape[1], -1))

    def forward(self, q, k, v):
        return self.attend(self.q_proj(q), self.k_proj(k), self.v_proj(v))


class TwoWayAttentionBlock(nn.Module):
    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False) -> None:
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.mlp.lin1.__class__ = CastCachedLinear                  # same parameters; the cast cache of the class above
        self.mlp.lin2.__class__ = CastCachedLinear
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe

    def token_side(self, queries, query_pe, k_img, v_img, t2i=None):
        """Steps (1)-(3) of the block given the image-side k / v of the token->image attention (already projected);
        t2i(q_projected) -> attention output before out_proj replaces the library attention when given."""
        if self.skip_first_layer_pe:
            queries = self.self_attn(q=queries, k=queries, v=queries)
        else:
            q = queries + query_pe
            queries = queries + self.self_attn(q=q, k=q, v=queries)
        queries = self.norm1(queries)
        ca = self.cross_attn_token_to_image
        qp = ca.q_proj(queries + query_pe)
        a = ca.attend(qp, k_img, v_img) if t2i is None else ca.out_proj(t2i(qp.float()))
        queries = self.norm2(queries + a)
        return self.norm3(queries + self.mlp(queries))

    def forward(self, queries, keys, query_pe, key_pe):
        ca = self.cross_attn_token_to_image
        queries = self.token_side(queries, query_pe, ca.k_proj(keys + key_pe), ca.v_proj(keys))
        q = queries + query_pe
        keys = self.norm4(keys + self.cross_attn_image_to_token(q=keys + key_pe, k=q, v=queries))
        return queries, keys


class TwoWayTransformer(nn.Module):
    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int,
                 activation: Type[nn.Module] = nn.ReLU, attention_downsample_rate: int = 2) -> None:
        super().__init__()
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate,
                                 skip_first_layer_pe=(i == 0)) for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)

    def forward_tokens(self, keys, key_pe, point_embedding):
        """Channels-last statement of forward(): keys, key_pe (B|1,N,C) image tokens, point_embedding (B,n,C)."""
        queries = point_embedding
        for layer in self.layers:
            queries, keys = layer(queries=queries, keys=keys, query_pe=point_embedding, key_pe=key_pe)
        q = queries + point_embedding
        queries = self.norm_final_attn(queries + self.final_attn_token_to_image(q=q, k=keys + key_pe, v=keys))
        return queries, keys

    def forward(self, image_embedding, image_pe, point_embedding):
        return self.forward_tokens(image_embedding.flatten(2).permute(0, 2, 1), image_pe.flatten(2).permute(0, 2, 1),
                                   point_embedding)


# ---- mask decoder ----------------------------------------------------------------------------------------------------
class MLP(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, sigmoid_output: bool = False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(CastCachedLinear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return torch.sigmoid(x) if self.sigmoid_output else x


class MaskDecoder(nn.Module):
    def __init__(self, *, transformer_dim: int, transformer: nn.Module, num_multimask_outputs: int = 3,
                 activation: Type[nn.Module] = nn.GELU, iou_head_depth: int = 3, iou_head_hidden_dim: int = 256) -> None:
        super().__init__()
        self.transformer_dim, self.transformer = transformer_dim, transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2),
            LayerNorm2d(transformer_dim // 4), activation(),
            nn.ConvTranspose2d(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2), activation())
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth)

    def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings,
                multimask_output: bool):
        masks, iou_pred = self.predict_masks(image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, sl, :, :], iou_pred[:, sl]

    # -- shared pieces ---------------------------------------------------------------------------------------------
    def _tokens(self, sparse):
        out = torch.cat([self.iou_token.weight, self.mask_tokens.weight], dim=0)
        return torch.cat((out.unsqueeze(0).expand(sparse.size(0), -1, -1), sparse.to(out.dtype)), dim=1)

    def _upscale(self, y0, b, h, w):
        """y0 (B, h*w, 4*C1): first transposed conv as a GEMM, columns ordered (c, dy, dx).  -> (B, 4h*4w, C2) channels
        last with pixel order (y, dy, dy2, x, dx, dx2) == row-major over the 4h x 4w grid."""
        ct1, ln, act1, ct2, act2 = self.output_upscaling
        c1, c2 = ct1.out_channels, ct2.out_channels
        u = y0.view(b, h, w, c1, 2, 2).permute(0, 1, 4, 2, 5, 3)                 # (b, h, dy, w, dx, c1)
        u = act1(F.layer_norm(u.float(), (c1,), ln.weight.float(), ln.bias.float(), ln.eps)).to(y0.dtype)
        w2 = ct2.weight.reshape(c1, c2 * 4).to(u.dtype)                           # (c1, (c2, dy2, dx2))
        v = act2(F.linear(u, w2.t(), ct2.bias.repeat_interleave(4).to(u.dtype)))  # (b, h, dy, w, dx, c2*4)
        v = v.view(b, h, 2, w, 2, c2, 2, 2).permute(0, 1, 2, 6, 3, 4, 7, 5)        # (b, h, dy, dy2, w, dx, dx2, c2)
        return v.reshape(b, 16 * h * w, c2)

    def _heads_out(self, hs, up, b, h, w):
        iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:1 + self.num_mask_tokens, :]
        hyper = torch.stack([self.output_hypernetworks_mlps[i](mask_toks[:, i, :]) for i in range(self.num_mask_tokens)], 1)
        masks = (hyper.to(up.dtype) @ up.transpose(1, 2)).view(b, -1, 4 * h, 4 * w)
        return masks.float(), self.iou_prediction_head(iou_tok).float()

    # -- the two paths ---------------------------------------------------------------------------------------------
    def predict_masks(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings):
        dt = _dtype() if image_embeddings.is_cuda else torch.float32
        shared = (image_embeddings.shape[0] == 1 and dense_prompt_embeddings.dim() == 4 and
                  dense_prompt_embeddings.stride(0) == 0 and dense_prompt_embeddings.stride(2) == 0 and
                  dense_prompt_embeddings.stride(3) == 0 and len(self.transformer.layers) == 2)
        with torch.autocast(device_type=image_embeddings.device.type, dtype=dt, enabled=dt != torch.float32):
            if shared and dt == torch.bfloat16 and self._fusable(image_embeddings, sparse_prompt_embeddings):
                return self._predict_masks_fused(image_embeddings, image_pe, sparse_prompt_embeddings,
                                                 dense_prompt_embeddings)
            if shared:
                return self._predict_masks_shared(image_embeddings, image_pe, sparse_prompt_embeddings,
                                                  dense_prompt_embeddings)
            return self._predict_masks_lib(image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings)

    def _predict_masks_lib(self, image_embeddings, image_pe, sparse, dense):
        """The reference's sequence (mask_decoder.py:106-143) on channels-last tokens."""
        tokens = self._tokens(sparse)
        B = tokens.shape[0]
        _, c, h, w = image_embeddings.shape
        keys = (torch.repeat_interleave(image_embeddings, B, dim=0) + dense).flatten(2).permute(0, 2, 1)
        hs, keys = self.transformer.forward_tokens(keys, image_pe.flatten(2).permute(0, 2, 1), tokens)
        ct1 = self.output_upscaling[0]
        w1 = ct1.weight.reshape(c, -1)                                            # (C, (c1, dy, dx))
        y0 = F.linear(keys, w1.t().to(keys.dtype), ct1.bias.repeat_interleave(4).to(keys.dtype))
        return self._heads_out(hs, self._upscale(y0, B, h, w), B, h, w)

    def _predict_masks_shared(self, image_embeddings, image_pe, sparse, dense):
        """Every prompt sees the same image tokens at the input of layer 0 (see the module docstring)."""
        tr = self.transformer
        L0, L1, fin = tr.layers[0], tr.layers[1], tr.final_attn_token_to_image
        tokens = self._tokens(sparse)
        B = tokens.shape[0]
        _, c, h, w = image_embeddings.shape
        keys0 = (image_embeddings + dense[:1]).flatten(2).permute(0, 2, 1)        # (1, N, C)
        pe = image_pe.flatten(2).permute(0, 2, 1)                                 # (1, N, C)
        kp0 = keys0 + pe
        # ---- layer 0: all image-side projections are per image -----------------------------------------------
        ca, ci = L0.cross_attn_token_to_image, L0.cross_attn_image_to_token
        queries = L0.token_side(tokens, tokens, ca.k_proj(kp0), ca.v_proj(keys0))
        keys1 = L0.norm4(keys0 + ci.attend(ci.q_proj(kp0), ci.k_proj(queries + tokens), ci.v_proj(queries)))   # (B,N,C)
        # ---- layer 1: one GEMM for [k | v | q] of the per-prompt image tokens; W pe terms per image -------------
        ca, ci = L1.cross_attn_token_to_image, L1.cross_attn_image_to_token
        d = ca.internal_dim
        wcat = torch.cat([ca.k_proj.weight, ca.v_proj.weight, ci.q_proj.weight], 0)
        bcat = torch.cat([ca.k_proj.bias, ca.v_proj.bias, ci.q_proj.bias], 0)
        kvq = F.linear(keys1, wcat.to(keys1.dtype), bcat.to(keys1.dtype))
        k_pe, q_pe = F.linear(pe, ca.k_proj.weight.to(pe.dtype)), F.linear(pe, ci.q_proj.weight.to(pe.dtype))
        queries = L1.token_side(queries, tokens, kvq[..., :d] + k_pe, kvq[..., d:2 * d])
        keys2 = L1.norm4(keys1 + ci.attend(kvq[..., 2 * d:] + q_pe, ci.k_proj(queries + tokens), ci.v_proj(queries)))
        # ---- final token->image attention + first transposed conv: one GEMM on keys2 --------------------------------
        ct1 = self.output_upscaling[0]
        wcat = torch.cat([fin.k_proj.weight, fin.v_proj.weight, ct1.weight.reshape(c, -1).t()], 0)
        bcat = torch.cat([fin.k_proj.bias, fin.v_proj.bias, ct1.bias.repeat_interleave(4)], 0)
        kvu = F.linear(keys2, wcat.to(keys2.dtype), bcat.to(keys2.dtype))
        k_pe = F.linear(pe, fin.k_proj.weight.to(pe.dtype))
        hs = tr.norm_final_attn(queries + fin.attend(fin.q_proj(queries + tokens), kvu[..., :d] + k_pe, kvu[..., d:2 * d]))
        return self._heads_out(hs, self._upscale(kvu[..., 2 * d:], B, h, w), B, h, w)

    # -- restructured path on the gfx950 kernels -------------------------------------------------------------------------
    def _fusable(self, emb, sparse):
        tr = self.transformer
        ci = tr.layers[0].cross_attn_image_to_token
        return policy.guard("sam.MaskDecoder", cuda=emb.is_cuda, vit_h_geometry=self.transformer_dim == 256 and tr.num_heads == 8 and ci.internal_dim == 128,
                            tokens_le_8=self.num_mask_tokens + 1 + sparse.shape[1] <= 8 and self.num_mask_tokens <= 4,
                            hw16=(emb.shape[2] * emb.shape[3]) % 16 == 0, have=ops.have("samdec_img2tok") and ops.have("samdec_upscale_heads"))

    def _prep(self, pe):
        """Weight-only (and positional-encoding-only) operands of the fused path, cached per weight version."""
        tr = self.transformer
        L1, fin = tr.layers[1], tr.final_attn_token_to_image
        ct1, ln, _, ct2, _ = self.output_upscaling
        ca_, ci_ = L1.cross_attn_token_to_image, L1.cross_attn_image_to_token
        srcs = [ca_.k_proj.weight, ca_.v_proj.weight, ci_.q_proj.weight, fin.k_proj.weight, fin.v_proj.weight, ct1.weight, ct2.weight,
                ln.weight, pe,
                # every bias the cached operands are made of as well (b_q1 / b_u / b_kvq / b_kvu / b2 / ln_b went stale when
                # only a bias was updated)
                ca_.k_proj.bias, ca_.v_proj.bias, ci_.q_proj.bias, fin.k_proj.bias, fin.v_proj.bias, ct1.bias, ct2.bias, ln.bias]
        key = tuple((t._version, t.data_ptr()) for t in srcs)
        c = getattr(self, "_s6d_prep", None)
        if c is None or c[0] != key:
            with torch.no_grad(), torch.autocast(device_type="cuda", enabled=False):
                bf = torch.bfloat16
                ca, ci = L1.cross_attn_token_to_image, L1.cross_attn_image_to_token
                C = self.transformer_dim
                pe32 = pe.float()
                d = dict(
                    w_kvq=torch.cat([ca.k_proj.weight, ca.v_proj.weight, ci.q_proj.weight], 0).to(bf).contiguous(),
                    b_kvq=torch.cat([ca.k_proj.bias, ca.v_proj.bias, ci.q_proj.bias], 0).to(bf).contiguous(),
                    kpe1=F.linear(pe32, ca.k_proj.weight.float()).to(bf).contiguous()[0],
                    qpe1=F.linear(pe32, ci.q_proj.weight.float()).to(bf).contiguous()[0],
                    # first transposed conv as GEMM rows ordered (dy, dx, c): each sub-pixel's 64 channels contiguous
                    w_kvu=torch.cat([fin.k_proj.weight, fin.v_proj.weight,
                                     ct1.weight.permute(2, 3, 1, 0).reshape(-1, C)], 0).to(bf).contiguous(),
                    b_kvu=torch.cat([fin.k_proj.bias, fin.v_proj.bias, ct1.bias.repeat(4)], 0).to(bf).contiguous(),
                    kpef=F.linear(pe32, fin.k_proj.weight.float()).to(bf).contiguous()[0],
                    # round 4 (raw token->image attention): only the image->token queries and the first transposed conv are still
                    # projected over the B x N image tokens
                    pe_bf=pe32.to(bf).contiguous()[0],
                    w_q1=ci.q_proj.weight.to(bf).contiguous(), b_q1=ci.q_proj.bias.to(bf).contiguous(),
                    w_u=ct1.weight.permute(2, 3, 1, 0).reshape(-1, C).to(bf).contiguous(), b_u=ct1.bias.repeat(4).to(bf).contiguous(),
                    w2t=ct2.weight.permute(2, 3, 1, 0).reshape(4 * ct2.out_channels, ct2.in_channels).to(bf).contiguous(),
                    b2=ct2.bias.float().contiguous(), ln_w=ln.weight.float().contiguous(), ln_b=ln.bias.float().contiguous())
            c = (key, d)
            self._s6d_prep = c
        return c[1]

    def _token_weights(self, li):
        """Operands of the token-side kernels (ops.samdec_tokens_pre / _post) for layer li: every Linear as (bf16 weight in
        matrix-instruction fragment order, f32 bias), the LayerNorms as (gamma, beta, eps); cached per weight version."""
        L = self.transformer.layers[li]
        sa, ca, ci = L.self_attn, L.cross_attn_token_to_image, L.cross_attn_image_to_token
        lins = [sa.q_proj, sa.k_proj, sa.v_proj, sa.out_proj, ca.q_proj, ca.out_proj, L.mlp.lin1, L.mlp.lin2, ci.k_proj, ci.v_proj]
        norms = [L.norm1, L.norm2, L.norm3]
        srcs = [t for m in lins + norms + [ca.k_proj, ca.v_proj, ci.q_proj, ci.out_proj] for t in (m.weight, m.bias)]
        key = tuple((t._version, t.data_ptr()) for t in srcs)
        cache = self.__dict__.setdefault("_s6d_tokw", {})
        c = cache.get(li)
        if c is None or c[0] != key:
            with torch.no_grad(), torch.autocast(device_type="cuda", enabled=False):
                bf = torch.bfloat16
                lw = [(ops.fragment_weight(m.weight.detach().to(bf).contiguous()), m.bias.detach().float().contiguous()) for m in lins]
                nw = [(m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous(), float(m.eps)) for m in norms]
                # the per-head folds around the two attention cores (ops.samdec_tokens_pre(fold=), ops.samdec_tokens_post(y=, expand=))
                fw = dict(wk=ops.fragment_weight(ca.k_proj.weight.detach().to(bf).t().contiguous()),            # (256,128) = W_k^T
                          wv=ops.fragment_weight(ca.v_proj.weight.detach().to(bf).contiguous()),                # (128,256)
                          bv=ca.v_proj.bias.detach().float().contiguous(),
                          wq=ops.fragment_weight(ci.q_proj.weight.detach().to(bf).t().contiguous()),            # (256,128) = W_q^T
                          bq=ci.q_proj.bias.detach().float().contiguous(),
                          wo=ops.fragment_weight(ci.out_proj.weight.detach().to(bf).contiguous()))              # (256,128)
            c = (key, lw, nw, fw)
            cache[li] = c
        return c[1], c[2], c[3]

    def _token_side(self, li, queries, tokens, t2i, x=None, pe_bf=None, fold_q=False):
        """TwoWayAttentionBlock.token_side of layer li + the k / v projections of its image->token attention, on the two
        token-side kernels (csrc/s6d_samtok.hip) around the token->image attention core -> (queries, extra); the module statements
        when the kernels do not apply (extra = None).
        With x (the image tokens the core attends, bf16) and pe_bf given, the core runs on the folded queries the first kernel
        writes, its W_v product and the operands of the image->token attention kernel (block-diagonal keys or their W_q fold,
        the values with out_proj folded in) are made by the second kernel: extra = dict(kexp | k256 + cb, vpt).  Otherwise `t2i`
        (projected queries -> attention output before out_proj) is called between the kernels: extra = (kt, vt)."""
        L = self.transformer.layers[li]
        if policy.guard("sam.MaskDecoder.token_side", cuda=queries.is_cuda, have=ops.have("samdec_tokens"), T8=queries.shape[1] <= 8,
                        f32=queries.dtype == torch.float32 and tokens.dtype == torch.float32, t2i=t2i is not None or x is not None,
                        geometry=L.self_attn.internal_dim == 256 and L.mlp.lin1.out_features == 2048):
            lw, nw, fw = self._token_weights(li)
            ca = L.cross_attn_token_to_image
            with torch.autocast(device_type="cuda", enabled=False):
                if x is not None and ops.have("samdec_tok2img_raw") and ops.have("samdec_token_folds"):    # (S6D_DISABLE_FUSED=samdec_token_folds: A/B)
                    sc = 1.4426950408889634 / math.sqrt(ca.internal_dim // ca.num_heads)
                    q1, qfold = ops.samdec_tokens_pre(queries, tokens, not L.skip_first_layer_pe, lw[0], lw[1], lw[2], lw[3], nw[0], lw[4],
                                                      fold=(fw["wk"], sc))
                    y = ops.samdec_tok2img_raw_core(qfold, x, pe_bf)
                    return ops.samdec_tokens_post(q1, None, tokens, lw[5], nw[1], lw[6], lw[7], nw[2], lw[8], lw[9], y=y,
                                                  vfold=(fw["wv"], fw["bv"]), expand=dict(wq=fw["wq"], bq=fw["bq"], wo=fw["wo"], fold_q=fold_q))
                q1, qp = ops.samdec_tokens_pre(queries, tokens, not L.skip_first_layer_pe, lw[0], lw[1], lw[2], lw[3], nw[0], lw[4])
                att = t2i(qp)
                q3, kt, vt = ops.samdec_tokens_post(q1, att, tokens, lw[5], nw[1], lw[6], lw[7], nw[2], lw[8], lw[9])
            return q3, (kt, vt)
        return L.token_side(queries, tokens, None, None, t2i), None

    @staticmethod
    def _rows_gemm(x, w, b):
        """x (B, N, K) bf16 @ w (n, K)^T + b over ALL B N image-token rows (1024 prompts x 4096 tokens = 4.2 M rows): the
        hand-written GEMM in row slabs below its 2 GiB operand limit when it applies (n % 128 == 0, K % 64 == 0;
        S6D_SAMDEC_GEMM=library opts out), else the library statement."""
        import os
        K, n = x.shape[-1], w.shape[0]
        if policy.guard("sam.MaskDecoder.rows_gemm", cuda=x.is_cuda, bf16=x.dtype == torch.bfloat16, N128=n % 128 == 0, K64=K % 64 == 0,
                        have=ops.have("gemm_bf16"), policy_kernel=policy.current().samdec_gemm != "library"):
            x2 = x.reshape(-1, K)
            M = x2.shape[0]
            out = torch.empty(M, n, dtype=torch.bfloat16, device=x.device)
            slab = max(256, ((1 << 30) // (2 * max(K, n))) // 256 * 256)            # rows per call: A and C slabs stay under 1 GiB
            bf = b.float()
            for r in range(0, M, slab):
                ops.gemm_bf16(x2[r:r + slab], w, bf, out=out[r:r + slab])
            return out.view(*x.shape[:-1], n)
        return F.linear(x, w, b)

    @staticmethod
    def _expand(att, queries, tokens, fold_q=False, ktvt=None):
        """Operands of s6d_samdec_img2tok_bf16 from the prompt tokens: block-diagonal scaled keys (B,64,128) and the
        values with out_proj folded in (B,256,64); slot j = head * 8 + token.
        fold_q: the image side's q projection folded into the keys (s6d_samdec_img2tok_raw_bf16) -> (kexp W_q (B,64,256) bf16,
        kexp . b_q (B,64) f32, vpt).
        ktvt: (k_proj(queries + tokens), v_proj(queries)) as (B,T,128) f32 when the token-side kernel has made them already."""
        B, T, _ = queries.shape
        H, hd = att.num_heads, att.internal_dim // att.num_heads
        if ktvt is not None:
            kt, vt = (ktvt[0] / math.sqrt(hd)).view(B, T, H, hd), ktvt[1].view(B, T, H, hd)
        else:
            kt = (att.k_proj(queries + tokens).float() / math.sqrt(hd)).view(B, T, H, hd)
            vt = att.v_proj(queries).float().view(B, T, H, hd)
        eye = torch.eye(H, device=kt.device, dtype=kt.dtype)
        kexp = F.pad(torch.einsum("bthd,hg->bhtgd", kt, eye), (0, 0, 0, 0, 0, 8 - T)).reshape(B, 8 * H, H * hd)
        wo = att.out_proj.weight.float().view(-1, H, hd)
        vpt = F.pad(torch.einsum("bthd,nhd->bnht", vt, wo), (0, 8 - T)).reshape(B, wo.shape[0], 8 * H)
        if fold_q:
            bf = torch.bfloat16
            with torch.autocast(device_type=kexp.device.type, enabled=False):
                # (B,64,128) @ (128,256) on the matrix cores: bf16 operands (what the unfolded path fed its score product), fp32
                # accumulation, bf16 result -- the float32 library GEMM cost 0.42 ms per 1024 prompts; the bias product stays float32
                k256 = kexp.to(bf) @ att.q_proj.weight.to(bf)
                cb = kexp.float() @ att.q_proj.bias.float()                                # (B,64)
            return k256.contiguous(), cb.contiguous(), vpt.to(bf).contiguous()
        return kexp.to(torch.bfloat16).contiguous(), vpt.to(torch.bfloat16).contiguous()

    def _predict_masks_fused(self, image_embeddings, image_pe, sparse, dense):
        """_predict_masks_shared with the per-prompt image-side work on the fused kernels: bf16 token tensors, one
        kernel per image->token attention block (incl. out_proj, residual, norm4), one for the whole output head."""
        tr = self.transformer
        L0, L1, fin = tr.layers[0], tr.layers[1], tr.final_attn_token_to_image
        bf = torch.bfloat16
        tokens = self._tokens(sparse)
        B, T = tokens.shape[0], tokens.shape[1]
        _, c, h, w = image_embeddings.shape
        keys0 = (image_embeddings + dense[:1]).flatten(2).permute(0, 2, 1)        # (1, N, C) fp32
        pe = image_pe.flatten(2).permute(0, 2, 1)
        P = self._prep(pe)
        kp0 = keys0 + pe
        # ---- layer 0 (image side shared by every prompt) -----------------------------------------------------------
        ca, ci = L0.cross_attn_token_to_image, L0.cross_attn_image_to_token
        t2i = ops.have("samdec_tok2img")
        # round 4: token->image attention on the RAW image tokens (k / v projections folded into the 8 x 8 queries, matrix cores):
        # no k / v tensor over the B x N image tokens is written or read (S6D_SAMDEC_T2I=kv: the round-3 form)
        import os
        raw = (ops.have("samdec_tok2img_raw") and (h * w) % 64 == 0 and policy.current().samdec_t2i == "raw")
        sc = 1.0 / math.sqrt(ca.internal_dim // ca.num_heads)
        keys0_bf = keys0.to(bf).contiguous()

        def t2i_raw(att, x, pe_):
            return lambda qp: ops.samdec_tok2img_raw(qp, x, pe_, att.k_proj.weight, att.v_proj.weight, att.v_proj.bias, sc)
        ktvt = None
        if raw:
            queries, ktvt = self._token_side(0, tokens.float(), tokens.float(), t2i_raw(ca, keys0_bf, P["pe_bf"]), x=keys0_bf, pe_bf=P["pe_bf"])
        elif t2i:
            kv0 = torch.cat([ca.k_proj(kp0), ca.v_proj(keys0)], -1).to(bf).contiguous()           # (1, N, 2d)
            queries = L0.token_side(tokens, tokens, None, None,
                                    lambda qp: ops.samdec_tok2img(qp, kv0, 0, ca.internal_dim, None, sc))
        else:
            queries = L0.token_side(tokens, tokens, ca.k_proj(kp0), ca.v_proj(keys0))
        if isinstance(ktvt, dict):
            kexp, vpt = ktvt["kexp"], ktvt["vpt"]
        else:
            kexp, vpt = self._expand(ci, queries, tokens, ktvt=ktvt)
        n4 = L0.norm4
        keys1 = ops.samdec_img2tok(ci.q_proj(kp0).to(bf).contiguous(), None, kexp, vpt, keys0_bf,
                                   ci.out_proj.bias.float(), n4.weight.float(), n4.bias.float(), n4.eps, T)
        # ---- layer 1 -----------------------------------------------------------------------------------------------
        ca, ci = L1.cross_attn_token_to_image, L1.cross_attn_image_to_token
        d = ca.internal_dim
        fold = raw and ops.have("samdec_img2tok_raw")
        ktvt = None
        if raw:
            queries, ktvt = self._token_side(1, queries.float(), tokens.float(), t2i_raw(ca, keys1, P["pe_bf"]), x=keys1, pe_bf=P["pe_bf"],
                                             fold_q=fold)
            q1 = None if fold else self._rows_gemm(keys1, P["w_q1"], P["b_q1"])    # (B, N, d) bf16: image->token queries only
        else:
            kvq = self._rows_gemm(keys1, P["w_kvq"], P["b_kvq"])                   # (B, N, 3d) bf16
            if t2i:
                queries = L1.token_side(queries, tokens, None, None,
                                        lambda qp: ops.samdec_tok2img(qp, kvq, 0, d, P["kpe1"], sc))
            else:
                queries = L1.token_side(queries, tokens, kvq[..., :d] + P["kpe1"], kvq[..., d:2 * d])
            q1 = kvq[..., 2 * d:]
        n4 = L1.norm4
        if fold:                                                                   # q projection folded into the expanded keys
            if isinstance(ktvt, dict):
                k256, cb, vpt = ktvt["k256"], ktvt["cb"], ktvt["vpt"]
            else:
                k256, cb, vpt = self._expand(ci, queries, tokens, fold_q=True, ktvt=ktvt)
            keys2 = ops.samdec_img2tok_raw(keys1, P["pe_bf"], k256, cb, vpt, keys1, ci.out_proj.bias.float(),
                                           n4.weight.float(), n4.bias.float(), n4.eps, T)
        else:
            if isinstance(ktvt, dict):
                kexp, vpt = ktvt["kexp"], ktvt["vpt"]
            else:
                kexp, vpt = self._expand(ci, queries, tokens, ktvt=ktvt)
            keys2 = ops.samdec_img2tok(q1, P["qpe1"], kexp, vpt, keys1, ci.out_proj.bias.float(),
                                       n4.weight.float(), n4.bias.float(), n4.eps, T)
        # ---- final token->image attention + output head ---------------------------------------------------------------
        qf = fin.q_proj(queries + tokens)
        if raw:
            a = fin.out_proj(t2i_raw(fin, keys2, P["pe_bf"])(qf.float()))
            up = self._rows_gemm(keys2, P["w_u"], P["b_u"])                        # (B, N, 4*c1) bf16: first transposed conv
        else:
            kvu = self._rows_gemm(keys2, P["w_kvu"], P["b_kvu"])                   # (B, N, 2d + 4*c1) bf16
            if t2i:
                a = fin.out_proj(ops.samdec_tok2img(qf.float(), kvu, 0, d, P["kpef"], sc))
            else:
                a = fin.attend(qf, kvu[..., :d] + P["kpef"], kvu[..., d:2 * d])
            up = kvu[..., 2 * d:]
        hs = tr.norm_final_attn(queries + a)
        iou_tok, mask_toks = hs[:, 0, :], hs[:, 1:1 + self.num_mask_tokens, :]
        hyper = torch.stack([self.output_hypernetworks_mlps[i](mask_toks[:, i, :]) for i in range(self.num_mask_tokens)], 1)
        masks = ops.samdec_upscale_heads(up, P["ln_w"], P["ln_b"], self.output_upscaling[1].eps, P["w2t"],
                                         P["b2"], hyper.float().contiguous(), h, w)
        return masks, self.iou_prediction_head(iou_tok).float()
