"""GPU parity of the fused MFMA attention (windowed / global, decomposed rel-pos bias) against the ORACLE's statement of the
same attention (oracle/sam.py windowed_attention_from_qkv: the code path of oracle.block, which tests/test_oracle_golden.py pins
to the reference's ImageEncoderViT), in fp32 on the same bf16-rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(B, H, nh, hd, ws, seed):
    g = torch.Generator().manual_seed(seed)
    S = ws if ws > 0 else H
    bias = 0.3 * torch.randn(3 * nh * hd, generator=g)
    rh = 0.3 * torch.randn(2 * S - 1, hd, generator=g)
    rw = 0.3 * torch.randn(2 * S - 1, hd, generator=g)
    qkv = torch.randn(B, H, H, 3 * nh * hd, generator=g)
    return bias, rh, rw, qkv


@pytest.mark.parametrize("B,H,nh,hd,ws", [(2, 32, 2, 80, 14), (1, 32, 2, 80, 0), (1, 64, 2, 80, 14),
                                          (1, 64, 1, 80, 0), (1, 64, 2, 64, 0), (2, 16, 3, 64, 7), (1, 16, 2, 64, 0), (1, 20, 1, 80, 14)])
def test_fused_attention_vs_oracle(B, H, nh, hd, ws):
    from oracle import sam as osam
    from sam6d_amd import ops
    assert ops.have("win_attention")
    bias, rh, rw, qkv = (t.to(torch.bfloat16) for t in _mk(B, H, nh, hd, ws, 1000 * H + ws + hd))
    out = ops.window_attention(qkv.cuda().contiguous(), bias.cuda().contiguous(), rh.cuda().contiguous(), rw.cuda().contiguous(), nh, ws,
                               hd ** -0.5).float().cpu()
    # the oracle in fp32 on the SAME bf16-rounded operands (padded window tokens carry the qkv bias: quirk Q2)
    ref = osam.windowed_attention_from_qkv(qkv.float(), bias.float(), rh.float(), rw.float(), nh, ws)
    err = (out - ref).abs()
    assert err.max() < 3e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("hd,grow,vscale", [(80, 30.0, 1.0), (80, 2.0, 1.0), (64, 30.0, 1.0), (80, 2.0, 2.0 ** 50)])
def test_global_attention_when_scores_outgrow_the_first_tile(hd, grow, vscale):
    """attn_global64_kernel keeps the FIRST key tile's row maximum for every later tile (process_tile_nomax, round 5) and repeats
    its tiles with the running-maximum arithmetic when a row sum leaves float32's comfortable range.  Image 1's queries get a
    +-2 pattern added and its keys from grid row 8 on ARE that pattern times `grow`: with 30 the later scores exceed the first
    tile's maximum by hundreds of log2 units (exp2 overflows -> the second pass must run), with 2 by some tens (P up to ~2^50,
    finite: the fast path must stay accurate).  Image 0 is ordinary, so one workgroup set takes the fallback and the other not.
    vscale = 2^50: image 1's VALUES are that large, so sum P V leaves float32 (2^90 x 2^50) on rows whose sum of P alone
    would have passed a 2^100 limit -- they must take the second pass (limit 2^60) and come out finite and accurate."""
    from oracle import sam as osam
    from sam6d_amd import ops
    H, nh = 64, 1
    bias, rh, rw, qkv = _mk(2, H, nh, hd, 0, 77)
    bias = torch.zeros_like(bias)
    pat = torch.where(torch.arange(hd) % 2 == 0, 1.0, -1.0)
    qkv[1, :, :, :hd] += 2 * pat
    qkv[1, 8:, :, hd:2 * hd] = grow * pat
    qkv[1, :, :, 2 * hd:] *= vscale
    bias, rh, rw, qkv = (t.to(torch.bfloat16) for t in (bias, rh, rw, qkv))
    # the premise, in log2 units: (largest late score) - (largest score against the first 64 keys), per query of image 1
    q, k = qkv[1, :, :, :hd].float().reshape(-1, hd), qkv[1, :, :, hd:2 * hd].float().reshape(-1, hd)
    sc = (q @ k.t()) * hd ** -0.5 * 1.4426950408889634
    gap = (sc[:, 512:].max(1).values - sc[:, :64].max(1).values)
    assert (gap.min() > 128) if grow == 30.0 else (20 < gap.min() and gap.max() < 90), (gap.min().item(), gap.max().item())
    out = ops.window_attention(qkv.cuda().contiguous(), bias.cuda().contiguous(), rh.cuda().contiguous(), rw.cuda().contiguous(), nh, 0,
                               hd ** -0.5).float().cpu()
    ref = osam.windowed_attention_from_qkv(qkv.float(), bias.float(), rh.float(), rw.float(), nh, 0)
    assert torch.isfinite(out).all()
    out[1] /= vscale
    ref[1] /= vscale
    err = (out - ref).abs()
    assert err.max() < 3e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("grow,vscale", [(30.0, 1.0), (2.0, 1.0), (2.0, 2.0 ** 50)])
def test_window_attention_when_scores_outgrow_the_first_key_rows(grow, vscale):
    """The persistent 14 x 14 window kernel streams its 32-key steps against the maximum of the window's FIRST TWO key rows
    (win16_pass_stream, round 5) and repeats the pass with that reference raised by 96 log2 units per round while a row sum is out of
    float32's comfortable range.  The keys of window rows >= 2 are a +-1 pattern times `grow`, the queries carry twice that pattern:
    with 30 the later scores exceed the reference by hundreds of log2 units (several retry rounds), with 2 by some tens (no retry, P
    up to ~2^50)."""
    from oracle import sam as osam
    from sam6d_amd import ops
    H, nh, hd, ws = 28, 2, 80, 14
    bias, rh, rw, qkv = _mk(2, H, nh, hd, ws, 91)
    bias = torch.zeros_like(bias)
    pat = torch.where(torch.arange(hd) % 2 == 0, 1.0, -1.0)
    late = (torch.arange(H) % ws) >= 2
    for h in range(nh):
        qkv[1, :, :, h * hd:(h + 1) * hd] += 2 * pat
        qkv[1, late, :, nh * hd + h * hd:nh * hd + (h + 1) * hd] = grow * pat
    qkv[1, :, :, 2 * nh * hd:] *= vscale                              # (large values under a large, in-range row sum)
    bias, rh, rw, qkv = (t.to(torch.bfloat16) for t in (bias, rh, rw, qkv))
    out = ops.window_attention(qkv.cuda().contiguous(), bias.cuda().contiguous(), rh.cuda().contiguous(), rw.cuda().contiguous(), nh, ws,
                               hd ** -0.5).float().cpu()
    ref = osam.windowed_attention_from_qkv(qkv.float(), bias.float(), rh.float(), rw.float(), nh, ws)
    assert torch.isfinite(out).all()
    out[1] /= vscale
    ref[1] /= vscale
    err = (out - ref).abs()
    assert err.max() < 3e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("ws", [14, 0])
def test_benched_launch_group_b16_h16(ws):
    """The launch group bench.py runs (16 frames x 16 heads x hd 80 on the 64 x 64 grid: 6400 window items through the
    persistent attn_window16p_kernel, 4096 workgroups of attn_global64_kernel, both with the XCD-aware item order that is
    taken when B * heads % 8 == 0): three distinct frames placed in the 16 slots in a scrambled order must come out, slot by
    slot, BIT FOR BIT as the same frame run alone (B = 1: a different item -> workgroup / XCD assignment, same arithmetic),
    and frame 0 is held to the oracle element by element."""
    from oracle import sam as osam
    from sam6d_amd import ops
    H, nh, hd = 64, 16, 80
    bias, rh, rw, qkv3 = (t.to(torch.bfloat16) for t in _mk(3, H, nh, hd, ws, 4242 + ws))
    order = [0, 2, 1, 1, 0, 2, 2, 0, 1, 0, 2, 1, 2, 2, 0, 1]
    dev = lambda t: t.cuda().contiguous()
    run = lambda x: ops.window_attention(dev(x), dev(bias), dev(rh), dev(rw), nh, ws, hd ** -0.5)
    alone = [run(qkv3[i:i + 1]) for i in range(3)]
    out = run(qkv3[order])
    assert out.shape == (16, H, H, nh * hd)
    for slot, i in enumerate(order):
        assert torch.equal(out[slot], alone[i][0]), f"slot {slot} (frame {i}) differs from the B = 1 launch"
    ref = osam.windowed_attention_from_qkv(qkv3[0:1].float(), bias.float(), rh.float(), rw.float(), nh, ws)
    err = (alone[0].float().cpu() - ref).abs()
    assert err.max() < 3e-2 and err.mean() < 3e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("B,H,nh,hd,ws", [(1, 28, 2, 80, 14), (1, 32, 2, 80, 0), (2, 20, 4, 80, 14), (1, 16, 2, 64, 7), (1, 64, 2, 80, 0)])
def test_head_major_layout_equals_token_major(B, H, nh, hd, ws):
    """The attention kernels on the head-major q/k/v tensor (3 heads, B H W, hd) -- what the qkv GEMM's column-block epilogue
    writes -- give bit for bit what they give on the Linear layout (B,H,W,3,heads,hd): only addresses differ."""
    from sam6d_amd import ops
    bias, rh, rw, qkv = (t.to(torch.bfloat16).cuda() for t in _mk(B, H, nh, hd, ws, 7 * H + ws + hd))
    tok = ops.window_attention(qkv.contiguous(), bias, rh.contiguous(), rw.contiguous(), nh, ws, hd ** -0.5)
    hm = qkv.view(B * H * H, 3 * nh, hd).permute(1, 0, 2).contiguous()
    out = ops.window_attention(hm, bias, rh.contiguous(), rw.contiguous(), nh, ws, hd ** -0.5, head_major_shape=(B, H, H))
    assert torch.equal(out, tok)


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


def test_seq_attention_float16_vs_torch(B=2, N=197, nh=12, hd=64):
    """The IEEE-half build of the sequence attention (PEM ViT-B shape) against the fp32 statement on the same half operands."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(N + nh)
    qkv = torch.randn(B, N, 3 * nh * hd, generator=g).to(torch.float16).cuda()
    out = ops.seq_attention(qkv, nh, hd ** -0.5)
    assert out.dtype == torch.float16
    q, k, v = qkv.float().cpu().view(B, N, 3, nh, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).transpose(1, 2).reshape(B, N, nh * hd)
    assert (out.float().cpu() - ref).abs().max() < 4e-3          # P is rounded to half (2^-11) before the PV product


def test_add_layernorm_float16():
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(300, 768, generator=g).to(torch.float16).cuda()
    d = torch.randn(300, 768, generator=g).to(torch.float16).cuda()
    w = (1 + 0.1 * torch.randn(768, generator=g)).cuda()
    b = (0.1 * torch.randn(768, generator=g)).cuda()
    xo, y = ops.add_layernorm(x, d, w, b, 1e-6)
    xr = (x.float() + d.float()).to(torch.float16)
    assert torch.equal(xo, xr)
    assert (y.float() - torch.nn.functional.layer_norm(xr.float(), (768,), w, b, 1e-6)).abs().max() < 4e-3


@pytest.mark.parametrize("B,N,nh,hd", [(3, 197, 12, 64), (2, 50, 2, 80), (1, 257, 4, 64)])
def test_seq_attention_vs_torch(B, N, nh, hd):
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
    out = ops.seq_attention(qkv, nh, hd ** -0.5).float()
    q, k, v = qkv.float().view(B, N, 3, nh, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).transpose(1, 2).reshape(B, N, nh * hd)
    assert (out - ref).abs().max() < 3e-2 and (out - ref).abs().mean() < 3e-3
    # head-major operand layout ((3 nh, B N, hd): the qkv GEMM's column-block output): the same values, bit for bit
    hm = qkv.view(B * N, 3 * nh, hd).transpose(0, 1).contiguous()
    assert torch.equal(ops.seq_attention(hm, nh, hd ** -0.5, seq_len=N).float(), out)
