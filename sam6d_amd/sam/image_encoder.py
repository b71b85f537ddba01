"""``ImageEncoderViT`` -- drop-in for segment_anything/modeling/image_encoder.py on MI355X.

Same constructor signature, attribute ``img_size``, forward (B,3,S,S) -> (B,out_chans,S/16,S/16)
and state_dict keys as the reference (image_encoder.py:17-116, common.py:13-43), so
``sam.load_state_dict(torch.load("sam_vit_h_4b8939.pth"))`` (build_sam.py:103-106) is unchanged.

Hardware-first execution:
  * compute dtype bf16 by default (``S6D_SAM_DTYPE`` = bf16 | fp32); the reference BOP run uses
    fp16 autocast (configs/machine/trainer/local.yaml:9), the custom demo fp32.
  * windows are never materialised: qkv is one GEMM over the 4096 real tokens (the reference pads
    64->70 first and runs the GEMM on 4900 tokens, image_encoder.py:168-174).  The padded tokens of
    the reference are zeros AFTER norm1, so their q/k/v equal the qkv bias; the fused attention
    kernel substitutes the bias for out-of-image keys (quirk Q2 preserved exactly) and writes
    straight into the (B,64,64,C) token map -- no partition / unpartition copies.
  * decomposed relative position bias is added inside the attention kernel from per-query tables
    (never a (heads,4096,4096) tensor).
"""
import os
from typing import Optional, Tuple, Type

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import policy
from ..utils.linear import fused_linear


def _dtype():
    return {"fp32": torch.float32, "bf16": torch.bfloat16}[policy.current().sam_dtype]


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim, mlp_dim, act=nn.GELU):
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)
        self.act = act()

    def forward(self, x):
        rows = int(policy.current().sam_mlp_rows)
        if rows <= 0 or x.numel() // x.shape[-1] <= rows:
            if isinstance(self.act, nn.GELU) and self.act.approximate == "none":
                # lin1 + bias + exact GELU in one GEMM epilogue, lin2 + bias in the other (common.py:13-28)
                return fused_linear(self.lin2, fused_linear(self.lin1, x, gelu=True))
            return self.lin2(self.act(self.lin1(x)))
        # Experiment knob (off by default, not yet measured): run lin1 -> GELU -> lin2 over row chunks so that a chunk's
        # (rows, mlp_dim) intermediate -- 168 MB at 16384 rows of ViT-H in bf16 -- can stay in the 256 MB Infinity Cache
        # between the three kernels instead of making two HBM round trips.
        flat = x.reshape(-1, x.shape[-1])
        out = torch.empty(flat.shape[0], self.lin2.out_features, dtype=flat.dtype, device=flat.device)
        exact = isinstance(self.act, nn.GELU) and self.act.approximate == "none"
        for a in range(0, flat.shape[0], rows):
            if exact:
                out[a:a + rows] = fused_linear(self.lin2, fused_linear(self.lin1, flat[a:a + rows], gelu=True))
            else:
                out[a:a + rows] = self.lin2(self.act(self.lin1(flat[a:a + rows])))
        return out.reshape(*x.shape[:-1], self.lin2.out_features)


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps

    def forward(self, x):  # channel LN on NCHW == layer_norm on NHWC
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class PatchEmbed(nn.Module):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans=3, embed_dim=768):
        super().__init__()
        assert tuple(kernel_size) == tuple(stride) and tuple(padding) == (0, 0)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)

    def forward(self, x):
        B, C, H, W = x.shape
        p = self.proj.kernel_size[0]
        if policy.guard("sam.PatchEmbed", cuda=x.is_cuda, half_dtype=x.dtype in (torch.bfloat16, torch.float16), p8=p % 8 == 0,
                        whole_patches=H % p == 0 and W % p == 0, have=ops.have("patchify")):
            x = ops.patchify(x.contiguous(), p)                   # one coalesced pass instead of a 6-d strided copy
        else:
            x = x.view(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, H // p, W // p, C * p * p)
        return fused_linear(self.proj, x, weight2d=self.proj.weight.flatten(1))


def _rel_table(size, rel_pos):
    """Rows of rel_pos used for a size x size grid: get_rel_pos (image_encoder.py:292-322) with
    q_size == k_size (always true here) reduces to the identity table when len == 2*size-1."""
    L = 2 * size - 1
    if rel_pos.shape[0] != L:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1).float(), size=L, mode="linear")
        rel_pos = r.reshape(-1, L).permute(1, 0).to(rel_pos.dtype)
    return rel_pos


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=True, use_rel_pos=False, rel_pos_zero_init=True,
                 input_size: Optional[Tuple[int, int]] = None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if use_rel_pos:
            assert input_size is not None
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))

    def _kernel_operands(self, S, dtype):
        """qkv bias and the two (2S-1, hd) position tables in the kernel's dtype, cached until a parameter changes (three casts
        per block and call otherwise: 192 small launches per 32-frame step)."""
        key = (S, dtype, self.qkv.bias._version, self.qkv.bias.data_ptr(), self.rel_pos_h._version, self.rel_pos_h.data_ptr(),
               self.rel_pos_w._version, self.rel_pos_w.data_ptr())
        c = getattr(self, "_s6d_attn_ops", None)
        if c is None or c[0] != key:
            c = (key, self.qkv.bias.detach().to(dtype).contiguous(), _rel_table(S, self.rel_pos_h.detach()).to(dtype).contiguous(),
                 _rel_table(S, self.rel_pos_w.detach()).to(dtype).contiguous())
            self._s6d_attn_ops = c
        return c[1], c[2], c[3]

    def forward(self, x, window_size=0, residual=None, qkv=None):
        """x: (B,H,W,C) token map (already normed).  window_size 0 = global attention.  residual: the block's shortcut -- the
        result is then shortcut + attention, with the add in the proj GEMM's epilogue (written over `residual`).  qkv: the
        (B,H,W,3C) projection when the caller has made it already (fp8 path: x is then only consulted for its shape)."""
        B, H, W, C = x.shape
        if qkv is not None:
            S = window_size if window_size > 0 else H
            bias, rh, rw = self._kernel_operands(S, qkv.dtype)
            out = ops.window_attention(qkv.contiguous(), bias, rh, rw, self.num_heads, window_size, self.scale)
            return fused_linear(self.proj, out, residual=residual)
        fused = policy.guard("sam.Attention", cuda=x.is_cuda, have=ops.have("win_attention"), bf16=x.dtype == torch.bfloat16, rel_pos=self.use_rel_pos)
        if fused:
            # S6D_QKV_LAYOUT=head: q / k / v head-major straight out of the GEMM's epilogue ((3 heads, B H W, hd): a head's rows of a
            # window row or of a key tile are contiguous whole lines for the attention kernels' fetches).  Measured neutral on the
            # step, so the default stays the Linear layout (fused_linear returns None)
            hm = fused_linear(self.qkv, x, col_block=C // self.num_heads)
            if hm is not None:
                S = window_size if window_size > 0 else H
                bias, rh, rw = self._kernel_operands(S, hm.dtype)
                out = ops.window_attention(hm, bias, rh, rw, self.num_heads, window_size, self.scale, head_major_shape=(B, H, W))
                return fused_linear(self.proj, out, residual=residual)
        qkv = fused_linear(self.qkv, x)                              # (B,H,W,3C): real tokens only
        if fused:
            S = window_size if window_size > 0 else H
            bias, rh, rw = self._kernel_operands(S, qkv.dtype)
            out = ops.window_attention(qkv.contiguous(), bias, rh, rw, self.num_heads, window_size, self.scale)
        else:
            out = self._attention_lib(qkv, B, H, W, C, window_size)
        return fused_linear(self.proj, out, residual=residual)

    def _attention_lib(self, qkv, B, H, W, C, ws):
        """Library-op statement of the same computation (device tensors; used when the fused
        kernel is absent from the loaded library or for fp32 runs)."""
        nh, hd = self.num_heads, C // self.num_heads
        if ws > 0:
            ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
            bias = self.qkv.bias.to(qkv.dtype) if self.qkv.bias is not None else qkv.new_zeros(3 * C)
            t = F.pad(qkv - bias, (0, 0, 0, pw, 0, ph)) + bias        # out-of-image tokens carry the bias
            Hp, Wp = H + ph, W + pw
            t = t.view(B, Hp // ws, ws, Wp // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3 * C)
            S = ws
        else:
            t = qkv.reshape(B, H * W, 3 * C)
            S = H
        Bw, N, _ = t.shape
        q, k, v = t.view(Bw, N, 3, nh, hd).permute(2, 0, 3, 1, 4).unbind(0)     # (Bw,nh,N,hd)
        attn = (q * self.scale) @ k.transpose(-1, -2)
        if self.use_rel_pos:
            Rh, Rw = _rel_table(S, self.rel_pos_h).to(q.dtype), _rel_table(S, self.rel_pos_w).to(q.dtype)
            idx = (torch.arange(S, device=q.device)[:, None] - torch.arange(S, device=q.device)[None, :]) + (S - 1)
            rq = q.reshape(Bw, nh, S, S, hd)
            rel_h = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh[idx])
            rel_w = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw[idx])
            attn = (attn.view(Bw, nh, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]).view(Bw, nh, N, N)
        o = (attn.softmax(dim=-1) @ v).transpose(1, 2).reshape(Bw, N, C)
        if ws > 0:
            o = o.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
            return o[:, :H, :W, :]
        return o.view(B, H, W, C)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer: Type[nn.Module] = nn.LayerNorm,
                 act_layer: Type[nn.Module] = nn.GELU, use_rel_pos=False, rel_pos_zero_init=True, window_size=0,
                 input_size: Optional[Tuple[int, int]] = None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.window_size = window_size

    def forward(self, x):
        x = x + self.attn(self.norm1(x), self.window_size)
        return x + self.mlp(self.norm2(x))


class ImageEncoderViT(nn.Module):
    def __init__(self, img_size: int = 1024, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768,
                 depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0, out_chans: int = 256,
                 qkv_bias: bool = True, norm_layer: Type[nn.Module] = nn.LayerNorm,
                 act_layer: Type[nn.Module] = nn.GELU, use_abs_pos: bool = True, use_rel_pos: bool = False,
                 rel_pos_zero_init: bool = True, window_size: int = 0,
                 global_attn_indexes: Tuple[int, ...] = ()) -> None:
        super().__init__()
        self.img_size = img_size
        self.patch_embed = PatchEmbed(kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size),
                                      in_chans=in_chans, embed_dim=embed_dim)
        self.pos_embed: Optional[nn.Parameter] = None
        if use_abs_pos:
            self.pos_embed = nn.Parameter(torch.zeros(1, img_size // patch_size, img_size // patch_size, embed_dim))
        self.blocks = nn.ModuleList()
        for i in range(depth):
            self.blocks.append(Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                                     rel_pos_zero_init=rel_pos_zero_init,
                                     window_size=window_size if i not in global_attn_indexes else 0,
                                     input_size=(img_size // patch_size, img_size // patch_size)))
        self.neck = nn.Sequential(
            nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
            nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False), LayerNorm2d(out_chans))

    def forward_tokens(self, x, upto=None):
        x = self.patch_embed(x)
        if self.pos_embed is not None:
            x = x + self.pos_embed.to(x.dtype)
        if policy.guard("sam.ImageEncoderViT.blocks", cuda=x.is_cuda, bf16=x.dtype == torch.bfloat16, have=ops.have("add_layernorm")):
            return self._blocks_fused(x, upto, own=True)       # x is this call's own tensor: updated in place
        for i, blk in enumerate(self.blocks):
            if upto is not None and i >= upto:
                break
            x = blk(x)
        return x

    @staticmethod
    def _ln_f32(norm):
        """fp32 copies of a LayerNorm's affine parameters, cached until the parameters change."""
        key = (norm.weight._version, norm.bias._version, norm.weight.data_ptr())
        c = getattr(norm, "_s6d_f32", None)
        if c is None or c[0] != key:
            c = (key, norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous())
            norm._s6d_f32 = c
        return c[1], c[2]

    def _blocks_fused(self, x, upto, own=False):
        """Same dataflow as Block.forward.  Round 3: the two residual adds of a block happen in the epilogues of the proj and
        lin2 GEMMs (s6d_gemm_bf16_res, in place on the stream tensor) and the LayerNorms are one-read passes; round 2 folded
        each add into the following LayerNorm pass (two reads + two writes per add: 6 % of the step).  Same arithmetic either
        way (the add rounds the GEMM's bf16 output + x to bf16): S6D_DISABLE_FUSED=gemm_bf16_res selects the round-2 form."""
        from ..utils.linear import lnfold_eligible, res_eligible
        x = x.contiguous()
        C = x.shape[-1]
        if _gemm_mode() in ("fp8", "fp8mx"):
            return self._blocks_fp8(x, upto, own)
        rows, hid = x.numel() // C, max(blk.mlp.lin1.out_features for blk in self.blocks)
        if policy.guard("sam.ImageEncoderViT.lnfold", cuda=x.is_cuda,
                        eligible=lnfold_eligible(x, C, C) and C % 32 == 0 and all(
                            lnfold_eligible(x, blk.mlp.lin1.out_features, C) and lnfold_eligible(x, C, blk.mlp.lin1.out_features)
                            and blk.attn.use_rel_pos for blk in self.blocks),
                        have=ops.have("win_attention"), one_launch_rows=rows <= ops.gemm_one_launch_rows(max(hid, 3 * C))):
            return self._blocks_lnfold(x, upto, own)
        if res_eligible(x, C, C) and all(res_eligible(x, C, blk.mlp.lin2.in_features) for blk in self.blocks):
            x = x.clone()                                        # the stream tensor is updated in place from here on
            for i, blk in enumerate(self.blocks):
                if upto is not None and i >= upto:
                    break
                g, b = self._ln_f32(blk.norm1)
                _, h = ops.add_layernorm(x, None, g, b, blk.norm1.eps)
                x = blk.attn(h, blk.window_size, residual=x)
                g, b = self._ln_f32(blk.norm2)
                _, h = ops.add_layernorm(x, None, g, b, blk.norm2.eps)
                x = fused_linear(blk.mlp.lin2, fused_linear(blk.mlp.lin1, h, gelu=True), residual=x)
            return x
        delta = None
        for i, blk in enumerate(self.blocks):
            if upto is not None and i >= upto:
                break
            g, b = self._ln_f32(blk.norm1)
            x, h = ops.add_layernorm(x, delta, g, b, blk.norm1.eps)
            a = blk.attn(h, blk.window_size)
            g, b = self._ln_f32(blk.norm2)
            x, h = ops.add_layernorm(x, a.contiguous(), g, b, blk.norm2.eps)
            delta = blk.mlp(h).contiguous()
        return x if delta is None else x + delta

    def _blocks_lnfold(self, x, upto, own=False):
        """Block.forward (image_encoder.py:166-182) with neither the residual adds nor the LayerNorms as passes of their own
        (round 2 / 3 spent 6 % of the step in add + LayerNorm: two reads and two writes of the token map per add):
          * `x + proj(..)` and `x + lin2(..)`: the residual tile rides through the matrix cores of the producing GEMM
            (s6d_gemm_bf16_res, in place on the stream tensor), whose epilogue also leaves per-row partial statistics of the new x;
          * `norm1(x)` / `norm2(x)`: s6d_ln_stats_finalize turns the partials into (mean, sigma) per token (a 20-MB pass), and the
            consuming GEMM (qkv, lin1 + GELU) multiplies the RAW stream by gamma * W, starting its accumulators at
            sigma b' - mean s and dividing by sigma in its epilogue (s6d_gemm_bf16_lnfold).  The normalised activations never exist.
        Out-of-image window tokens keep the ORIGINAL qkv bias (the reference pads after norm1: zeros through qkv = its bias)."""
        from ..utils.linear import _cached, lnfold_cached
        if not own:
            x = x.clone()                                        # the stream tensor is updated in place from here on
        B, H, W, C = x.shape
        M = B * H * W
        x2 = x.view(M, C)
        blocks = list(self.blocks)[:upto] if upto is not None else list(self.blocks)
        if not blocks:
            return x
        st = ops.row_stats(x2, blocks[0].norm1.eps)
        sp = torch.empty(C // 32, 2, M, dtype=torch.float32, device=x.device)
        for i, blk in enumerate(blocks):
            at = blk.attn
            wf, cs, bf = lnfold_cached(at.qkv, blk.norm1)
            qkv = ops.gemm_bf16_lnfold(x2, st, wf, cs, bf).view(B, H, W, 3 * C)
            S = blk.window_size if blk.window_size > 0 else H
            bias, rh, rw = at._kernel_operands(S, qkv.dtype)
            a = ops.window_attention(qkv, bias, rh, rw, at.num_heads, blk.window_size, at.scale)
            wp, bp = _cached(at.proj, at.proj.weight)
            ops.gemm_bf16(a.reshape(M, C), wp, bp, residual=x2, out=x2, stats_partial=sp)
            st = ops.ln_stats_finalize(sp, 32, blk.norm2.eps)
            w1, c1, b1 = lnfold_cached(blk.mlp.lin1, blk.norm2)
            h = ops.gemm_bf16_lnfold(x2, st, w1, c1, b1, gelu=True)
            w2, b2 = _cached(blk.mlp.lin2, blk.mlp.lin2.weight)
            last = i + 1 == len(blocks)
            ops.gemm_bf16(h, w2, b2, residual=x2, out=x2, stats_partial=None if last else sp)
            if not last:
                st = ops.ln_stats_finalize(sp, 32, blocks[i + 1].norm1.eps)
        return x

    def _blocks_fp8(self, x, upto, own=False):
        """BASELINE configs[4] (never the headline): the two LayerNorm-fed GEMMs of every block -- qkv (1280 -> 3840) and lin1
        (1280 -> 5120, + GELU), 58 % of the encoder's GEMM FLOP -- on the fp8 matrix cores.  LayerNorm writes its output as e4m3
        bytes with one power-of-two scale per token (s6d_layernorm_fp8: no extra pass), the weights carry one power-of-two scale
        per output channel (utils/fp8.py), both ride in the matrix instruction's block-scale operands (s6d_gemm_fp8).  proj and
        lin2 (their inputs come out of the attention kernel / the GELU epilogue in bf16, and a per-token scale needs the whole
        row) stay bf16, as do attention, the residual stream and the neck."""
        from ..utils import fp8
        C = x.shape[-1]
        if not (ops.have("gemm_fp8") and ops.have("layernorm_fp8") and x.is_cuda and x.dtype == torch.bfloat16 and C % 256 == 0):
            raise RuntimeError("S6D_SAM_GEMM=fp8 needs the fp8 kernels of libsam6d_hip.so, a bf16 device token map and C % 256 == 0")
        from ..utils.linear import _cached
        B, H, W, _ = x.shape
        M = B * H * W
        if ops.have("gemm_bf16_res") and all(blk.attn.use_rel_pos for blk in self.blocks) and ops.have("win_attention") \
                and M <= ops.gemm_one_launch_rows(max(blk.mlp.lin1.out_features for blk in self.blocks)):
            # the residual adds in the bf16 proj / lin2 GEMMs (accumulators start at bias + residual, in place on the stream
            # tensor): the quantising LayerNorm then reads ONE tensor and writes its e4m3 rows
            if not own:
                x = x.clone()
            x2 = x.view(M, C)
            mx = _gemm_mode() == "fp8mx" and ops.have("gemm_fp8_mx") and M % 256 == 0
            delta = None
            for i, blk in enumerate(self.blocks):
                if upto is not None and i >= upto:
                    break
                at = blk.attn
                g, b = self._ln_f32(blk.norm1)
                if delta is None:
                    h8, hs = ops.layernorm_fp8(x, g, b, blk.norm1.eps)
                else:                                              # fp8mx: the previous block's lin2 output joins the stream here
                    x, h8, hs = ops.layernorm_fp8(x, g, b, blk.norm1.eps, delta=delta)
                    x2 = x.view(M, C)
                wq, ws, bq = fp8.cached_weight(at.qkv)
                qkv = ops.gemm_fp8(h8, hs, wq, ws, bq).view(B, H, W, 3 * C)
                S = blk.window_size if blk.window_size > 0 else H
                bias, rh, rw = at._kernel_operands(S, qkv.dtype)
                a = ops.window_attention(qkv, bias, rh, rw, at.num_heads, blk.window_size, at.scale)
                wp, bp = _cached(at.proj, at.proj.weight)
                ops.gemm_bf16(a.reshape(M, C), wp, bp, residual=x2, out=x2)
                g, b = self._ln_f32(blk.norm2)
                h8, hs = ops.layernorm_fp8(x, g, b, blk.norm2.eps)
                w1, s1, b1 = fp8.cached_weight(blk.mlp.lin1)
                if mx:
                    # S6D_SAM_GEMM=fp8mx (round 4): lin1's GELU epilogue writes e4m3 with MX block scales (one E8M0 byte per token
                    # and 32 channels), lin2 multiplies them on the fp8 matrix cores -- the 5120-wide hidden activations never exist
                    # in bf16.  lin2's output is a bf16 delta that the next quantising LayerNorm adds to the stream
                    # (s6d_add_layernorm_fp8): the fp8 kernel has no registers left for the residual epilogue.
                    q8, qs = ops.gemm_fp8_gelu_mx(h8, hs, w1, s1, b1)
                    w2q, w2s, b2f = fp8.cached_weight(blk.mlp.lin2)
                    delta = ops.gemm_fp8_mxa(q8, qs, w2q, w2s, b2f).view(x.shape)
                else:
                    w2, b2 = _cached(blk.mlp.lin2, blk.mlp.lin2.weight)
                    ops.gemm_bf16(ops.gemm_fp8(h8, hs, w1, s1, b1, gelu=True), w2, b2, residual=x2, out=x2)
            return x if delta is None else x + delta
        delta = None
        for i, blk in enumerate(self.blocks):
            if upto is not None and i >= upto:
                break
            # each residual add folded into the following quantising LayerNorm pass, as in the round-2 bf16 loop
            g, b = self._ln_f32(blk.norm1)
            if delta is None:
                h8, hs = ops.layernorm_fp8(x, g, b, blk.norm1.eps)
            else:
                x, h8, hs = ops.layernorm_fp8(x, g, b, blk.norm1.eps, delta=delta)
            wq, ws, bq = fp8.cached_weight(blk.attn.qkv)
            a = blk.attn(x, blk.window_size, qkv=ops.gemm_fp8(h8, hs, wq, ws, bq))
            g, b = self._ln_f32(blk.norm2)
            x, h8, hs = ops.layernorm_fp8(x, g, b, blk.norm2.eps, delta=a.contiguous())
            w1, s1, b1 = fp8.cached_weight(blk.mlp.lin1)
            delta = fused_linear(blk.mlp.lin2, ops.gemm_fp8(h8, hs, w1, s1, b1, gelu=True)).contiguous()
        return x if delta is None else x + delta

    def neck_nhwc(self, t):
        """neck (image_encoder.py:90-104) on the (B,H,W,C) token map, channels-last throughout:
        1x1 conv = GEMM, LayerNorm2d = LN over the last dim, 3x3 conv = 9 shifted GEMMs accumulated
        (MIOpen's bf16 NHWC 3x3 falls back to a naive kernel: 29 ms/call measured, profiles/r01_sam8_*)."""
        c1, n1, c3, n2 = self.neck[0], self.neck[1], self.neck[2], self.neck[3]
        B, H, W, C = t.shape
        dt = t.dtype
        y = fused_linear(c1, t, weight2d=c1.weight.flatten(1))
        kern = policy.guard("sam.neck.LayerNorm2d", cuda=y.is_cuda, bf16=dt == torch.bfloat16,
                            have=ops.have("add_layernorm") and ops.have("layernorm_f32out"), C8=y.shape[-1] % 8 == 0)
        if kern:                                                            # LayerNorm2d = LN over the channel (last) dim: one kernel pass
            g, b = self._ln_f32(n1)
            y = ops.add_layernorm(y.contiguous(), None, g, b, n1.eps)[1]
        else:
            y = F.layer_norm(y.float(), (y.shape[-1],), n1.weight.float(), n1.bias.float(), n1.eps).to(dt)
        Co = y.shape[-1]
        if policy.guard("sam.neck.conv3x3", cuda=y.is_cuda, bf16=dt == torch.bfloat16, have=ops.have("gemm_bf16"), K64=(9 * Co) % 64 == 0,
                        N128=c3.weight.shape[0] % 128 == 0):
            # the 3x3 convolution as ONE GEMM over K = 9 Ci (the nine shifted views side by side, weight in (dy, dx, ci) order):
            # fp32 accumulation over all taps inside the kernel instead of nine bf16 partial products summed in fp32 passes
            if ops.have("im2col3x3") and Co % 8 == 0:
                cols = ops.im2col3x3(y.contiguous())
            else:
                yp = F.pad(y, (0, 0, 1, 1, 1, 1))                           # zero pad H and W by 1
                cols = torch.cat([yp[:, dy:dy + H, dx:dx + W, :] for dy in range(3) for dx in range(3)], dim=-1)
            acc = fused_linear(c3, cols, weight2d=c3.weight.permute(0, 2, 3, 1).reshape(c3.weight.shape[0], -1))
            if kern:                                                        # bf16 GEMM output -> fp32 embedding, statistics in fp32
                g, b = self._ln_f32(n2)
                return ops.layernorm_f32out(acc.contiguous(), g, b, n2.eps).permute(0, 3, 1, 2)
            acc = acc.float()
        else:
            yp = F.pad(y, (0, 0, 1, 1, 1, 1))                               # zero pad H and W by 1
            w = c3.weight.to(dt)                                            # (Co, Ci, 3, 3)
            acc = None
            for dy in range(3):
                for dx in range(3):
                    part = F.linear(yp[:, dy:dy + H, dx:dx + W, :], w[:, :, dy, dx])
                    acc = part.float() if acc is None else acc + part.float()
            if c3.bias is not None:
                acc = acc + c3.bias.float()
        z = F.layer_norm(acc, (Co,), n2.weight.float(), n2.bias.float(), n2.eps)
        return z.permute(0, 3, 1, 2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = _dtype()
        in_dtype = x.dtype
        if dt != torch.float32 and x.is_cuda:
            with torch.autocast(device_type="cuda", dtype=dt):
                t = self.forward_tokens(x.to(dt))
            return self.neck_nhwc(t.to(dt)).to(in_dtype)
        return self.neck_nhwc(self.forward_tokens(x)).to(in_dtype)


def _gemm_mode():
    """S6D_SAM_GEMM = bf16 (default, BASELINE configs[1]) | fp8 (configs[4]: qkv and lin1 on the fp8 matrix cores) | fp8mx (round 4:
    lin2 too, fed by lin1's MX-scaled e4m3 output)."""
    return policy.current().sam_gemm


PIXEL_MEAN = (123.675, 116.28, 103.53)      # build_sam.py:95-96
PIXEL_STD = (58.395, 57.12, 57.375)


def preprocess(x, img_size=1024, pixel_mean=PIXEL_MEAN, pixel_std=PIXEL_STD, out_dtype=None):
    """Drop-in for ``Sam.preprocess`` (segment_anything/modeling/sam.py:164-174): normalise colours and zero-pad
    to a square input, fused with the cast to the encoder's compute dtype.  x: (B,3,h,w) or (3,h,w) float."""
    squeeze = x.dim() == 3
    if squeeze:
        x = x[None]
    out_dtype = out_dtype or _dtype()
    y = ops.sam_preprocess(x.float().contiguous(), pixel_mean, pixel_std, img_size, out_dtype)
    return y[0] if squeeze else y


def build_vit_h():
    """ViT-H configuration of segment_anything/build_sam.py:14-21,55-80."""
    from functools import partial
    return ImageEncoderViT(depth=32, embed_dim=1280, img_size=1024, mlp_ratio=4,
                           norm_layer=partial(nn.LayerNorm, eps=1e-6), num_heads=16, patch_size=16, qkv_bias=True,
                           use_rel_pos=True, global_attn_indexes=(7, 15, 23, 31), window_size=14, out_chans=256)
