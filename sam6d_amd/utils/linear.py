"""nn.Linear (+ exact GELU) through the hand-written bf16 GEMM (csrc/s6d_gemm.hip) for the ViTs on the hot path.

The modules keep their nn.Linear parameters (state_dict surface of the reference unchanged); this helper only decides
how the statement  act(x @ W^T + b)  is executed: on a device bf16 activation with N % 128 == 0 and K % 64 == 0 it is one
launch of s6d_gemm_bf16 (bias and GELU in the epilogue; `residual=`: the block's residual add in the epilogue too), otherwise
the library statement.  `S6D_DISABLE_FUSED=gemm_bf16`
turns the kernel off (A/B runs)."""
import os

import torch
import torch.nn.functional as F

from .. import ops
from .. import policy


def _cached(lin, w2d, dtype=torch.bfloat16):
    """bf16 (or float16) weight (N,K) + fp32 bias of a Linear / 1x1-or-patch Conv, cached until the parameters change."""
    b = lin.bias
    key = (lin.weight._version, lin.weight.data_ptr(), lin.weight.dtype, None if b is None else (b._version, b.data_ptr()), dtype)
    c = getattr(lin, "_s6d_gemm", None)
    if c is None or c[0] != key:
        wb = w2d.detach().to(dtype).contiguous()
        bf = None if b is None else b.detach().float().contiguous()
        c = (key, wb, bf)
        lin._s6d_gemm = c
    return c[1], c[2]


class CastCachedLinear(torch.nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose autocast casts are cached ACROSS calls.  Under torch.autocast every
    parameter is re-cast in every autocast region; for the mask decoder's token side -- some forty small Linear layers per batch of
    prompts -- that is two 5-us cast kernels per layer and call (120 per frame, profiles/r04_library_ops_frame.txt).  Here the cast
    copies are kept until the parameter changes (version / storage) and the product runs on them with autocast off: the same
    operands, the same library GEMM, the same bits.  Inference only (no_grad); anything else is nn.Linear.forward."""

    def forward(self, x):
        if x.is_cuda and torch.is_autocast_enabled("cuda") and not torch.is_grad_enabled():
            dt = torch.get_autocast_dtype("cuda")
            w, b = self.weight, self.bias
            key = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()), dt)
            c = self.__dict__.get("_s6d_cast")
            if c is None or c[0] != key:
                c = (key, w.detach().to(dt), None if b is None else b.detach().to(dt))
                self.__dict__["_s6d_cast"] = c
            with torch.autocast(device_type="cuda", enabled=False):
                return F.linear(x if x.dtype == dt else x.to(dt), c[1], c[2])
        return super().forward(x)


def eligible(x, n_out, k_in):
    if x.is_cuda and x.dtype == torch.float16:           # IEEE half (PEM ViT-B): the 256 x 256-tile kernel only
        return n_out % 256 == 0 and k_in % 64 == 0 and ops.have("gemm_f16")
    return (x.is_cuda and x.dtype == torch.bfloat16 and n_out % 128 == 0 and k_in % 64 == 0 and ops.have("gemm_bf16"))


def res_eligible(x, n_out, k_in):
    if x.dtype != torch.bfloat16:
        return False
    return _res_eligible(x, n_out, k_in)


def _res_eligible(x, n_out, k_in):
    """x + Linear(a) in one launch (s6d_gemm_bf16_res): the 256 x 256-tile kernel only.  On its own (the LayerNorm still a pass that
    reads the sum back) it buys nothing -- one tensor read moves from the add into the GEMM -- so it is OFF unless S6D_GEMM_RES=1;
    the ViT-H block loop uses it together with the folded LayerNorm (lnfold_eligible below), where both passes disappear
    (profiles/r03_lnfold.txt)."""
    return (eligible(x, n_out, k_in) and n_out % 256 == 0 and ops.have("gemm_bf16_res")
            and policy.current().gemm_res == "1")


def lnfold_weights(weight, bias, gamma, beta):
    """LayerNorm folded into the Linear behind it:  LN(x) W^T + b = rstd (x W'^T - mean s) + b'  with
    W' = bf16(gamma * W),  s = row sums of W' (of the ROUNDED weights: the kernel multiplies those),  b' = b + W beta.
    -> (W' bf16 (N,K), s f32 (N), b' f32 (N)); sums in float64."""
    w = weight.detach().double()
    wf = (w * gamma.detach().double()[None, :]).to(torch.bfloat16).contiguous()
    cs = wf.double().sum(1).float().contiguous()
    b0 = 0.0 if bias is None else bias.detach().double()
    bf = (b0 + w @ beta.detach().double()).float().contiguous()
    return wf, cs, bf


def lnfold_cached(lin, norm):
    """lnfold_weights(lin, norm), cached on `lin` until one of the four parameters changes."""
    ps = (lin.weight, lin.bias, norm.weight, norm.bias)
    key = tuple(None if t is None else (t._version, t.data_ptr(), t.dtype) for t in ps)
    c = getattr(lin, "_s6d_lnfold", None)
    if c is None or c[0] != key:
        c = (key,) + lnfold_weights(lin.weight, lin.bias, norm.weight, norm.bias)
        lin._s6d_lnfold = c
    return c[1], c[2], c[3]


def lnfold_eligible(x, n_out, k_in):
    """The folded residual + LayerNorm block loop (s6d_gemm_bf16_res with row statistics -> s6d_gemm_bf16_lnfold): bf16 stream,
    every Linear of the block on the 256 x 256-tile kernel.  S6D_LNFOLD=0 turns it off (A/B runs)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and n_out % 256 == 0 and k_in % 64 == 0 and ops.have("gemm_bf16_lnfold")
            and policy.current().lnfold != "0" and "gemm_bf16" not in policy.current().disable_fused)


def fused_linear(lin, x, gelu=False, weight2d=None, col_block=0, residual=None):
    """act(lin(x)); `weight2d` overrides lin.weight for conv weights viewed as (N, K).  col_block > 0 (kernel path only, N % 256
    == 0, S6D_QKV_LAYOUT=head): the result comes back as (N / col_block, M, col_block) -- the head-major q/k/v layout of the attention
    kernels -- or None when that path does not apply (the caller then takes the plain form).  Off by default: measured on the
    ViT-H shapes (pass r2s) the attention kernels gain 2 % from whole-line fetches, the qkv GEMM loses 2.5 % on its scattered
    column-block stores, the step is unchanged (143.2 vs 144.4 frames/s)."""
    w = lin.weight if weight2d is None else weight2d
    N, K = w.shape
    if col_block:
        if eligible(x, N, K) and N % 256 == 0 and policy.current().qkv_layout == "head":
            wb, bf = _cached(lin, w)
            return ops.gemm_bf16(x, wb, bf, gelu=gelu, col_block=col_block)
        return None
    if residual is not None:
        # residual + lin(x): in the GEMM's epilogue when the kernel takes the shape, written over the residual (the caller's
        # stream tensor: nothing else holds it), otherwise the two statements
        if res_eligible(x, N, K) and residual.dtype == torch.bfloat16 and residual.is_contiguous():
            wb, bf = _cached(lin, w)
            return ops.gemm_bf16(x, wb, bf, residual=residual, out=residual.reshape(-1, N)).reshape(residual.shape)
        return residual + fused_linear(lin, x, gelu=gelu, weight2d=weight2d)
    if eligible(x, N, K):
        wb, bf = _cached(lin, w, x.dtype)
        return ops.gemm_bf16(x, wb, bf, gelu=gelu)
    # the library statement (rocBLAS / hipBLASLt on the device): recorded, and an error under strict mode (sam6d_amd/policy.py)
    policy.guard("utils.fused_linear", cuda=x.is_cuda, half_dtype=x.dtype in (torch.bfloat16, torch.float16),
                 shape=x.dtype == torch.float16 and N % 256 == 0 and K % 64 == 0 or x.dtype == torch.bfloat16 and N % 128 == 0 and K % 64 == 0,
                 have=ops.have("gemm_bf16"))
    y = F.linear(x, w.to(x.dtype), None if lin.bias is None else lin.bias.to(x.dtype))
    return F.gelu(y) if gelu else y
