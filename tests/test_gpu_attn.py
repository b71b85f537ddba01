"""GPU parity of the fused MFMA attention (windowed / global, decomposed rel-pos bias) against the
fp32 library-op statement of the same attention, itself pinned to the reference through
tests/test_host_sam.py (mini encoder vs reference golden)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(B, H, nh, hd, ws, seed):
    from sam6d_amd.sam.image_encoder import Attention
    g = torch.Generator().manual_seed(seed)
    S = ws if ws > 0 else H
    att = Attention(nh * hd, num_heads=nh, qkv_bias=True, use_rel_pos=True, input_size=(S, S)).eval()
    with torch.no_grad():
        att.qkv.bias.copy_(0.3 * torch.randn(3 * nh * hd, generator=g))
        att.rel_pos_h.copy_(0.3 * torch.randn(2 * S - 1, hd, generator=g))
        att.rel_pos_w.copy_(0.3 * torch.randn(2 * S - 1, hd, generator=g))
    qkv = torch.randn(B, H, H, 3 * nh * hd, generator=g)
    return att, qkv


@pytest.mark.parametrize("B,H,nh,hd,ws", [(2, 32, 2, 80, 14), (1, 32, 2, 80, 0), (1, 64, 2, 80, 14),
                                          (1, 64, 1, 80, 0), (2, 16, 3, 64, 7), (1, 16, 2, 64, 0), (1, 20, 1, 80, 14)])
def test_fused_attention_vs_library_statement(B, H, nh, hd, ws):
    from sam6d_amd import ops
    assert ops.have("win_attention")
    att, qkv = _mk(B, H, nh, hd, ws, 1000 * H + ws + hd)
    att = att.cuda()
    qkv_bf = qkv.cuda().to(torch.bfloat16)
    bias_bf = att.qkv.bias.detach().to(torch.bfloat16)
    rh = att.rel_pos_h.detach().to(torch.bfloat16).contiguous()
    rw = att.rel_pos_w.detach().to(torch.bfloat16).contiguous()
    out = ops.window_attention(qkv_bf.contiguous(), bias_bf.contiguous(), rh, rw, nh, ws, att.scale).float()
    # fp32 statement on the SAME bf16-rounded operands
    with torch.no_grad():
        att.qkv.bias.copy_(bias_bf.float())
        att.rel_pos_h.copy_(rh.float())
        att.rel_pos_w.copy_(rw.float())
        ref = att._attention_lib(qkv_bf.float(), B, H, H, nh * hd, ws)
    err = (out - ref).abs()
    assert err.max() < 3e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())


def test_no_bias_variant_matches_sdpa():
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, nh, hd = 2, 14, 3, 64
    qkv = torch.randn(B, H, H, 3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
    bias = torch.zeros(3 * nh * hd, dtype=torch.bfloat16, device="cuda")
    out = ops.window_attention(qkv, bias, None, None, nh, 0, hd ** -0.5).float()
    q, k, v = qkv.float().view(B, H * H, 3, nh, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v
    ref = ref.transpose(1, 2).reshape(B, H, H, nh * hd)
    assert (out - ref).abs().max() < 3e-2


@pytest.mark.parametrize("rows,C", [(1000, 1280), (37, 160), (5, 768), (64, 2048)])
def test_add_layernorm_bf16(rows, C):
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).cuda().to(torch.bfloat16)
    d = torch.randn(rows, C, generator=g).cuda().to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    b = (0.1 * torch.randn(C, generator=g)).cuda()
    xo, y = ops.add_layernorm(x, d, w, b, 1e-6)
    xr = (x.float() + d.float()).to(torch.bfloat16)
    assert torch.equal(xo, xr)
    yr = torch.nn.functional.layer_norm(xr.float(), (C,), w, b, 1e-6)
    assert (y.float() - yr).abs().max() < 2e-2
    x2, y2 = ops.add_layernorm(x, None, w, b, 1e-6)
    assert x2 is x
    assert (y2.float() - torch.nn.functional.layer_norm(x.float(), (C,), w, b, 1e-6)).abs().max() < 2e-2


@pytest.mark.parametrize("B,N,nh,hd", [(3, 197, 12, 64), (2, 50, 2, 80), (1, 257, 4, 64)])
def test_seq_attention_vs_torch(B, N, nh, hd):
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
    out = ops.seq_attention(qkv, nh, hd ** -0.5).float()
    q, k, v = qkv.float().view(B, N, 3, nh, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).transpose(1, 2).reshape(B, N, nh * hd)
    assert (out - ref).abs().max() < 3e-2 and (out - ref).abs().mean() < 3e-3
