"""Fused fp32 Linear of the PEM point transformer (s6d_linear_f32: split-bf16 matrix cores + bias / ReLU / residual / LayerNorm
epilogue) against the library statements in float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # M, K, N, bias, relu, residual, layernorm
    (100, 256, 256, True, False, False, False),       # ragged M (two row tiles, one partial)
    (64, 256, 256, False, False, False, False),
    (197, 256, 512, True, True, False, False),        # AttentionOutput.expand + ReLU (two column tiles)
    (197, 512, 256, True, False, True, True),         # AttentionOutput.squeeze + residual + norm (K = 512)
    (300, 256, 768, True, False, False, False),       # q | k | v in one launch
    (130, 256, 256, True, False, True, True),         # linear + residual + norm
    (70, 256, 256, True, True, True, False),
]


@pytest.mark.parametrize("M,K,N,bias,relu,res,ln", CASES)
def test_linear_f32_vs_library(M, K, N, bias, relu, res, ln):
    """fp32-class result: the 3-term split leaves ~2^-17 relative per product, i.e. errors of order 1e-5 of sum |x||w| (the
    library's own fp32 GEMM sits at 1e-6); after LayerNorm the bound is relative to the normalised scale."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g) * 0.7
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g) if res else None
    gm, bt = (1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)) if ln else (None, None)
    hi, lo = ops.split_weight(w.cuda())
    assert (hi.float().cpu() + lo.float().cpu() - w).abs().max() <= 2.0 ** -16 * w.abs().max()
    y = ops.linear_f32(x.cuda(), hi, lo, None if b is None else b.cuda(), relu=relu, residual=None if r is None else r.cuda(),
                       ln=None if not ln else (gm.cuda(), bt.cuda(), 1e-5)).cpu()
    ref = x.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if relu:
        ref = ref.relu()
    if r is not None:
        ref = ref + r.double()
    scale = (x.abs().double() @ w.abs().double().t())
    tol = 2e-5 * scale + 1e-6
    if ln:
        pre = ref
        ref = torch.nn.functional.layer_norm(pre, (N,), gm.double(), bt.double(), 1e-5)
        tol = (2e-5 * scale.amax(1, keepdim=True) / pre.std(1, keepdim=True)) * (1 + gm.abs().double()) + 1e-5
    err = (y.double() - ref).abs()
    assert (err <= tol).all(), (err / tol).max().item()
    assert y.shape == (M, N) and y.dtype == torch.float32


def test_linear_f32_at_the_dense_stage_shape():
    """65536 rows x 256 -> 256 (the fine stage's per-point layers at B = 32): a row sample against float64."""
    from sam6d_amd import ops
    if not torch.cuda.is_available():
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(9)
    M, K, N = 65536, 256, 256
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 16
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    gm, bt = torch.ones(N), torch.zeros(N)
    hi, lo = ops.split_weight(w.cuda())
    y = ops.linear_f32(x.cuda(), hi, lo, b.cuda(), residual=r.cuda(), ln=(gm.cuda(), bt.cuda(), 1e-5)).cpu()
    rows = torch.randperm(M, generator=g)[:2048]
    ref = torch.nn.functional.layer_norm(x[rows].double() @ w.double().t() + b.double() + r[rows].double(), (N,), gm.double(), bt.double(), 1e-5)
    assert (y[rows].double() - ref).abs().max() < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,J", [(2, 100, 37), (3, 64, 196), (32, 2048, 196)])
def test_linear_attention_vs_the_library_statement(B, I, J):
    """s6d_linear_attention_f32 (focus map of q + k^T v + q . sum k + (q kv) z + head merge, csrc/s6d_linattn.hip) against the
    statement it replaces (LinearAttention.forward's library branch = transformer.py:536-564) in float64, with k | v as the strided
    halves of one projection output, ragged I (rows past I are not stored)."""
    import torch.nn.functional as F

    from sam6d_amd import ops
    if not torch.cuda.is_available() and B * I > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(B + I + J)
    xq = torch.randn(B, I, 256, generator=g).cuda()
    kvp = torch.randn(B, J, 512, generator=g).cuda()                  # k | v of one projection launch
    scale = (0.3 * torch.randn(256, generator=g)).cuda()
    inv = 1.0 / F.softplus(scale)

    def focus(t, p=3):
        t = (F.relu(t) + 1e-6) * inv.to(t.dtype)
        n = t.norm(dim=-1, keepdim=True)
        t = t ** p
        return t / t.norm(dim=-1, keepdim=True) * n

    kf = ops.linear_attn_focus(kvp[..., :256].contiguous(), inv, 3)
    out = ops.linear_attention(xq, inv, 3, kf, kvp[..., 256:])
    sentinel = torch.full((B, I + 1, 256), 7.0).cuda()                # nothing is written past row I of a batch element
    assert out.shape == (B, I, 256) and torch.isfinite(out).all() and sentinel[0, I, 0] == 7.0

    def split(t):
        return t.view(t.shape[0], t.shape[1], 4, 64).transpose(1, 2)
    q, k, v = split(focus(xq.double())), split(focus(kvp[..., :256].double())), split(kvp[..., 256:].double())
    z = 1.0 / (q @ k.sum(dim=2).unsqueeze(-1) + 1e-6)
    ref = ((q @ (k.transpose(-1, -2) @ v)) * z).transpose(1, 2).reshape(B, I, 256)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-5 * (1 + ref.abs().max().item()), err


@pytest.mark.parametrize("kind,B,N", [("rpe", 2, 197), ("mha", 3, 61), ("linear", 2, 300), ("rpe", 1, 32), ("linear", 5, 2047)])   # (the last one: more than 8192 rows = the 64-row form)
def test_attention_output_chain_equals_the_three_launches_bit_for_bit(kind, B, N):
    """csrc/s6d_pchain.hip (round 6): norm(linear(att) + x) -> AttentionOutput as ONE kernel over 32-row strips against the three
    s6d_linear_f32 launches it replaces (S6D_DISABLE_FUSED=attn_output_chain) on the real layer classes: equal bits (the same
    3-term products in the same order, the same fixed-order LayerNorm sums), ragged row counts included; and within fp32-class
    distance of the library statements in float64."""
    from sam6d_amd import ops, policy
    from sam6d_amd.pem import layers as L
    from sam6d_amd.utils import seeded
    assert ops.have("attn_output_chain")
    g = torch.Generator().manual_seed(B * 1000 + N)
    layer = {"rpe": L.RPETransformerLayer, "mha": L.TransformerLayer, "linear": L.LinearTransformerLayer}[kind](256).eval()
    seeded.load_seeded(layer, 5)
    layer = layer.cuda()
    x = torch.randn(B, N, 256, generator=g).cuda()
    mem = torch.randn(B, 77, 256, generator=g).cuda()
    emb = torch.randn(B, N, N, 256, generator=g).cuda() * 0.1
    arg = emb if kind == "rpe" else mem
    with torch.no_grad():
        policy.reset_library_branch_hits()
        fused = layer(x, arg)
        assert not policy.library_branch_hits(), policy.library_branch_hits()
        with policy.use(disable_fused="attn_output_chain"):
            three = layer(x, arg)
        assert ("pem.attention_output_chain", "have") in policy.library_branch_hits()
        assert torch.equal(fused, three), float((fused - three).abs().max())
        # float64 statement of the chain on the kernel's own attention output
        att = layer.attention.attention(x, arg) if kind != "mha" else layer.attention.attention(x, arg, arg)
        a, o = layer.attention, layer.output
        d = lambda t: t.double()      # noqa: E731
        h = torch.nn.functional.layer_norm(d(x) + d(att) @ d(a.linear.weight).t() + d(a.linear.bias), (256,), d(a.norm.weight), d(a.norm.bias), a.norm.eps)
        e = torch.relu(h @ d(o.expand.weight).t() + d(o.expand.bias))
        y = torch.nn.functional.layer_norm(h + e @ d(o.squeeze.weight).t() + d(o.squeeze.bias), (256,), d(o.norm.weight), d(o.norm.bias), o.norm.eps)
    assert (fused.double() - y).abs().max().item() <= 2e-4 * max(1.0, y.abs().max().item())
