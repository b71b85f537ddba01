"""GPU parity of the pose-solver kernels vs the CPU oracle (oracle/pem.py)."""
import pytest
import torch

from oracle import pem as opem
from sam6d_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from sam6d_amd import ops
    return ops


def test_rot_from_h_vs_svd(ops):
    g = torch.Generator().manual_seed(0)
    H = torch.randn(4096, 3, 3, generator=g)
    U, _, V = torch.svd(H.double())
    eye = torch.eye(3, dtype=torch.float64).repeat(len(H), 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ U.transpose(1, 2)))
    ref = (V @ eye @ U.transpose(1, 2)).float()
    out = ops.rot_from_h(H.cuda()).cpu()
    s = torch.linalg.svdvals(H.double())
    well = (s[:, 1] - s[:, 2] > 1e-3 * s[:, 0])
    assert (out - ref)[well].abs().max() < 1e-5


@pytest.mark.parametrize("B,N", [(5, 2048), (2, 197), (1, 7)])
def test_weighted_procrustes_vs_oracle_and_batch_invariance(ops, B, N):
    """The fused weighted_procrustes of compute_fine_Rt (model_utils.py:268-271 -> :287-363) against the oracle's restatement on
    a noisy rigid pair with a third of the weights zero (the fine stage's row sums are), and the property it was written for:
    an instance's (R, t) is bit-identical whether it is solved alone or in a batch."""
    from oracle import pem as opem
    g = torch.Generator().manual_seed(3)
    src = torch.randn(B, N, 3, generator=g) * 0.3
    Rg = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    Rg = Rg * torch.sign(torch.linalg.det(Rg)).view(-1, 1, 1)
    ref = src @ Rg.transpose(1, 2) + torch.randn(B, 1, 3, generator=g) * 0.2 + 0.01 * torch.randn(B, N, 3, generator=g)
    w = torch.rand(B, N, generator=g) * (torch.rand(B, N, generator=g) > 0.33)
    R, t = ops.weighted_procrustes(src.cuda(), ref.cuda(), w.cuda(), 0.0, 1e-5)
    Ro, to = opem.weighted_procrustes(src, ref, w, 0.0)
    assert (R.cpu() - Ro).abs().max() < 2e-6 and (t.cpu() - to).abs().max() < 2e-6, ((R.cpu() - Ro).abs().max(), (t.cpu() - to).abs().max())
    if N >= 197:
        assert (R.cpu() - Rg).abs().max() < 0.05                               # ref ~ src R^T + t: the planted rotation comes back
    if N >= 197:                                                               # thresholded weights (the coarse call's form); with 7 points
        Rt, tt = ops.weighted_procrustes(src.cuda(), ref.cuda(), w.cuda(), 0.5, 1e-5)      # two would be left: no unique rotation
        Rto, tto = opem.weighted_procrustes(src, ref, w, 0.5)
        assert (Rt.cpu() - Rto).abs().max() < 2e-6 and (tt.cpu() - tto).abs().max() < 2e-6
    for b in range(B):
        R1, t1 = ops.weighted_procrustes(src[b:b + 1].contiguous().cuda(), ref[b:b + 1].contiguous().cuda(), w[b:b + 1].contiguous().cuda(), 0.0, 1e-5)
        assert torch.equal(R1[0], R[b]) and torch.equal(t1[0], t[b])


def test_half_stored_geo_embedding_and_its_reader(ops):
    """S6D_PEM_GEO_DTYPE=fp16: s6d_geo_embedding_f16 stores the SAME values rounded to IEEE half (relative 2^-11), and the RPE
    attention core on the half tensor equals the core on that tensor widened to float32 bit for bit (it widens in registers)."""
    from sam6d_amd.pem.layers import GeometricStructureEmbedding
    from sam6d_amd.pem.pose_estimation_model import default_cfg
    from sam6d_amd.utils import seeded
    geo = seeded.load_seeded(GeometricStructureEmbedding(default_cfg().geo_embedding).eval(), 4).cuda()
    pts = synth.pem_inputs(2, seed=9, n_pts=37, with_rgb=False)["pts"].cuda() * 5
    d_idx, a_idx = geo.get_embedding_indices(pts)
    idx4 = torch.cat([d_idx.unsqueeze(-1), a_idx], dim=-1).contiguous()
    args = (idx4, geo.proj_d.weight.contiguous(), geo.proj_d.bias, geo.proj_a.weight.contiguous(), geo.proj_a.bias, geo.embedding.div_term.contiguous())
    e32 = ops.geo_embedding(*args)
    e16 = ops.geo_embedding(*args, out_dtype=torch.float16)
    assert e16.dtype == torch.float16 and torch.equal(e16.cpu(), e32.cpu().half())
    # pre-split weights (what the module passes since round 5): the same bits as the in-kernel split, in both storage types
    split = geo._split_weights()
    assert split is not None and split[0].shape == (2, 256, 256) and split[0].dtype == torch.bfloat16
    assert torch.equal(ops.geo_embedding(*args, split=split).cpu(), e32.cpu())
    assert torch.equal(ops.geo_embedding(*args, out_dtype=torch.float16, split=split).cpu(), e16.cpu())
    # ... and both kernels that serve the pre-split operands (round 6: geo_embed2_kernel builds the sinusoid fragments in registers)
    try:
        ops.set_geo_embed_form(2)
        assert torch.equal(ops.geo_embedding(*args, split=split).cpu(), e32.cpu())
        assert torch.equal(ops.geo_embedding(*args, out_dtype=torch.float16, split=split).cpu(), e16.cpu())
    finally:
        ops.set_geo_embed_form(1)
    g = torch.Generator().manual_seed(2)
    B, N = pts.shape[:2]
    q, k, v = (torch.randn(B, N, 256, generator=g).cuda() for _ in range(3))
    qt, qb = torch.randn(B, 4, N, 256, generator=g).cuda() * 0.1, torch.randn(B, 4, N, generator=g).cuda()
    a = ops.rpe_attention(q, k, v, qt, qb, e16, 0.125)
    b = ops.rpe_attention(q, k, v, qt, qb, e16.float(), 0.125)
    assert torch.equal(a.cpu(), b.cpu())


def test_pose_hypotheses_vs_oracle(ops):
    B, N, n = 3, 196, 6000
    inp = synth.pem_inputs(B, seed=5, n_pts=N, with_rgb=False)
    p1, p2 = inp["pts"] * 5, inp["dense_po"] * 5
    g = torch.Generator().manual_seed(1)
    pair = torch.randint(0, N * N, (B, 3 * n), generator=g)
    pair[:, :5] = N * N  # searchsorted can return the end bin: exercises the clamps
    i1 = torch.clamp(pair.div(N, rounding_mode="floor"), max=N - 1)
    i2 = torch.clamp(pair % N, max=N - 1)
    a = torch.gather(p1, 1, i1.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n, 3, 3)
    b = torch.gather(p2, 1, i2.unsqueeze(2).expand(-1, -1, 3)).reshape(B * n, 3, 3)
    Rr, tr = opem.weighted_procrustes(b, a, None, weight_thresh=0.5)
    dr = torch.norm((a - tr.unsqueeze(1)) @ Rr - b, dim=2).mean(1)
    R, t, d = ops.pose_hypotheses(p1.cuda(), p2.cuda(), pair.int().cuda())
    R, t, d = R.cpu().reshape(-1, 3, 3), t.cpu().reshape(-1, 3), d.cpu().reshape(-1)
    H = (b - b.mean(1, keepdim=True)).transpose(1, 2) @ (a - a.mean(1, keepdim=True))
    s = torch.linalg.svdvals(H.double())
    well = (s[:, 1] > 1e-3 * s[:, 0]) & (s[:, 0] - s[:, 1] > 1e-3 * s[:, 0])   # non-degenerate triangles
    assert well.float().mean() > 0.9
    assert (R - Rr)[well].abs().max() < 2e-3 and (R - Rr)[well].abs().mean() < 1e-5
    assert (d - dr)[well].abs().max() < 1e-3
    assert (t - tr)[well].abs().max() < 5e-3


@pytest.mark.parametrize("B,M,n_u", [(3, 197, 18000), (2, 41, 900), (2, 9, 300), (3, 21, 600)])   # the last two: fewer rows than the per-wave scratch
def test_coarse_sample_vs_oracle(ops, B, M, n_u):
    """Sampling head of compute_coarse_Rt (model_utils.py:203-219) in one kernel against the reference statements: labels exact;
    sampled bins identical except where a uniform falls within float32 rounding of a bin boundary (the bins differ in their
    last bits: hardware exp and s * sqrt(s) against libm's exp and pow; the prefix sums are accumulated in float64 on both sides):
    at most 0.2 % of the draws, each landing on a bin whose cumulative value is within 2e-5 of the reference's (the hardware
    exponential is good to ~2e-6 relative, and a bin's cumulative value carries the drift of all bins before it)."""
    g = torch.Generator().manual_seed(M)
    f1 = torch.nn.functional.normalize(torch.randn(B, M, 64, generator=g), dim=2)
    f2 = torch.nn.functional.normalize(f1[:, torch.randperm(M, generator=g)] + 0.3 * torch.randn(B, M, 64, generator=g), dim=2)
    atten = f1 @ f2.transpose(1, 2) / 0.1
    u = torch.rand(B, n_u, generator=g)
    score, w1, _ = opem.soft_assignment(atten)
    sc = score.reshape(B, -1) ** 1.5
    cum = torch.cumsum(sc, 1)
    cum = cum / (cum[:, -1].unsqueeze(1) + 1e-8)
    ref = torch.searchsorted(cum, u)
    pair, w = ops.coarse_sample(atten.cuda(), u.cuda())
    pair, w = pair.cpu().long(), w.cpu()
    assert torch.equal(w, w1)
    diff = pair != ref
    assert diff.float().mean() <= 2e-3, diff.float().mean()
    if diff.any():                                                     # a differing draw sits on a boundary: same cumulative value
        b = diff.nonzero()[:, 0]
        got, want = cum[b, pair[diff].clamp(max=cum.shape[1] - 1)], cum[b, ref[diff].clamp(max=cum.shape[1] - 1)]
        assert (got - want).abs().max() < 2e-5


@pytest.mark.parametrize("B,N,n1,n2", [(2, 196, 6000, 300), (2, 40, 300, 30)])
def test_coarse_Rt_kernel_chain_vs_oracle(ops, B, N, n1, n2):
    """solvers.coarse_Rt through its five kernels (sampling head, hypotheses, smallest-k, min-dist, scored arg-max) against the
    oracle's compute_coarse_Rt on a known-answer case: same hypothesis => same pose."""
    from sam6d_amd.pem import solvers
    g = torch.Generator().manual_seed(N + n1)
    p2 = torch.randn(B, N, 3, generator=g) * 0.4
    Rgt = synth.random_rotations(B, g)
    tgt = 0.1 * torch.randn(B, 3, generator=g)
    p1 = p2 @ Rgt.transpose(1, 2) + tgt[:, None, :]
    f = torch.nn.functional.normalize(torch.randn(B, N + 1, 32, generator=g), dim=2)
    atten = f @ torch.nn.functional.normalize(f + 0.2 * torch.randn(B, N + 1, 32, generator=g), dim=2).transpose(1, 2) / 0.1
    model = p2[:, : max(N // 2, 16)].contiguous()
    u = torch.rand(B, 3 * n1, generator=g)
    Rr, tr = opem.coarse_Rt(atten, p1, p2, model, u, n1, n2)
    R, t = solvers.coarse_Rt(atten.cuda(), p1.cuda(), p2.cuda(), model.cuda(), u.cuda(), n1, n2)
    assert (R.cpu() - Rr).norm(dim=(1, 2)).max() < 1e-3 and (t.cpu() - tr).abs().max() < 1e-4
    assert (R.cpu() - Rgt).norm(dim=(1, 2)).max() < 1e-2


def test_smallest_k_and_hypothesis_select_vs_library(ops):
    """topk(largest=False) + gathers, and the scored arg-max over hypotheses (model_utils.py:233-246), as kernels."""
    g = torch.Generator().manual_seed(3)
    B, n, k, N = 3, 6000, 300, 196
    dis = torch.rand(B, n, generator=g)
    dis[0, 100] = dis[0, 7]                                            # a tie: the lower index first
    Rs = synth.random_rotations(B * n, g).reshape(B, n, 3, 3)
    ts = torch.randn(B, n, 3, generator=g)
    Rk, tk, idx = (t.cpu() for t in ops.smallest_k(dis.cuda(), Rs.cuda(), ts.cuda(), k))
    val, _ = torch.topk(dis, k, dim=1, largest=False)
    assert torch.equal(torch.gather(dis, 1, idx.long()), val)          # ascending values (indices may differ only on ties)
    assert (idx[0] == 7).nonzero().item() + 1 == (idx[0] == 100).nonzero().item() if (idx[0] == 7).any() and (idx[0] == 100).any() else True
    ar = torch.arange(B)[:, None]
    assert torch.equal(Rk, Rs[ar, idx.long()]) and torch.equal(tk, ts[ar, idx.long()])
    dmin = torch.rand(B, k, N, generator=g)
    w1 = (torch.rand(B, N, generator=g) > 0.3).float()
    w1[2] = 0                                                          # no foreground point: every score is 0 -> hypothesis 0
    sc = w1.sum(1, keepdim=True) / ((dmin * w1.unsqueeze(1)).sum(2) + 1e-8)
    best = sc.max(1)[1]
    R, t = (x.cpu() for x in ops.hypothesis_select(dmin.cuda(), w1.cuda(), Rk.cuda(), tk.cuda()))
    assert torch.equal(R, Rk[torch.arange(B), best]) and torch.equal(t, tk[torch.arange(B), best])


@pytest.mark.parametrize("N,P,Nm", [(196, 300, 1024), (2048, 1, 1024), (70, 3, 1021), (33, 2, 5)])
def test_min_dist_vs_oracle(ops, N, P, Nm):
    """(the model cloud is read four points per trip since round 6: 1021 and 5 points exercise the padded tail)"""
    B = 2
    g = torch.Generator().manual_seed(N)
    pts = torch.randn(B, N, 3, generator=g)
    model = torch.randn(B, Nm, 3, generator=g)
    R = synth.random_rotations(B * P, g).reshape(B, P, 3, 3)
    t = 0.1 * torch.randn(B, P, 3, generator=g)
    tp = ((pts.unsqueeze(1) - t.unsqueeze(2)) @ R).double()
    ref = torch.cdist(tp.reshape(B * P, N, 3), model.double().repeat_interleave(P, 0)).min(2)[0].reshape(B, P, N)
    out = ops.min_dist(pts.cuda(), R.cuda(), t.cuda(), model.cuda()).cpu()
    assert (out.double() - ref).abs().max() < 1e-5


@pytest.mark.parametrize("B,N", [(2, 197), (3, 50)])
@pytest.mark.parametrize("fold", ["1", "0"])
def test_rpe_attention_vs_oracle(ops, B, N, fold):
    """Fused RPE attention (q~.e rewrite, streamed embedding) vs the reference formulation.  fold = 1 (round 4, default): W_p folded
    into the q | k | v projection, the attention core reads q~ and qb in place from that launch's output (s6d_rpe_attention_packed_f32);
    fold = 0: the `W_p^T q` products as library einsums in front of s6d_rpe_attention_strided_f32."""
    import os

    from sam6d_amd.pem.layers import RPEMultiHeadAttention
    from sam6d_amd.utils import seeded
    os.environ["S6D_RPE_FOLD"] = fold; __import__("sam6d_amd.policy").policy.reload()
    calls = []
    real = ops.rpe_attention_packed
    ops.rpe_attention_packed = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        _rpe_case(ops, B, N)
    finally:
        ops.rpe_attention_packed = real
        os.environ.pop("S6D_RPE_FOLD", None); __import__("sam6d_amd.policy").policy.reload()
    assert len(calls) == (1 if fold == "1" else 0)


def _rpe_case(ops, B, N):
    from sam6d_amd.pem.layers import RPEMultiHeadAttention
    from sam6d_amd.utils import seeded
    m = RPEMultiHeadAttention(256).eval()
    seeded.load_seeded(m, 4)
    W = {"a." + k: v for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B * 100 + N)
    x = torch.randn(B, N, 256, generator=g)
    emb = 0.5 * torch.randn(B, N, N, 256, generator=g)
    with torch.no_grad():
        ref = opem.rpe_attention(W, "a", x, emb)
        assert ops.have("rpe_attention")
        out = m.cuda()(x.cuda(), emb.cuda()).cpu()
    assert (out - ref).abs().max() < 2e-5, (out - ref).abs().max()


@pytest.mark.parametrize("B,N", [(2, 197), (1, 37)])
def test_geo_embedding_vs_oracle(ops, B, N):
    """Fused sincos -> split-bf16 MFMA -> max kernel vs the fp32 reference formulation."""
    from sam6d_amd.pem.layers import GeometricStructureEmbedding
    from sam6d_amd.pem.pose_estimation_model import default_cfg
    from sam6d_amd.utils import seeded
    m = GeometricStructureEmbedding(default_cfg().geo_embedding).eval()
    seeded.load_seeded(m, 2)
    W = {"geo_embedding." + k: v for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    pts = torch.randn(B, N, 3, generator=g) * 0.5
    pts[:, 0] = 100.0                                   # the background point of pose_estimation_model.py:27
    with torch.no_grad():
        ref = opem.geo_embedding(W, pts)
        assert ops.have("geo_embedding")
        out = m.cuda()(pts.cuda()).cpu()
    # entry (0,0) is the reference's own noise floor (sqrt of a cancelling x^2-2xy+y^2 at |x|^2 = 3e4)
    mask = torch.ones(B, N, N, dtype=torch.bool)
    mask[:, 0, 0] = False
    err = (out - ref).abs()[mask]
    assert err.max() < 2e-3 and err.mean() < 2e-5, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("B,M", [(2, 2049), (3, 300), (1, 17)])
def test_fine_assign_vs_oracle(ops, B, M):
    """3-pass fused dual-softmax / masking / normalised assignment vs the reference chain."""
    g = torch.Generator().manual_seed(M)
    f1 = torch.nn.functional.normalize(torch.randn(B, M, 64, generator=g), dim=2)
    f2 = torch.nn.functional.normalize(f1[:, torch.randperm(M, generator=g)] + 0.3 * torch.randn(B, M, 64, generator=g), dim=2)
    atten = f1 @ f2.transpose(1, 2) / 0.1
    pts2 = torch.randn(B, M - 1, 3, generator=g)
    amat, w1, _ = opem.soft_assignment(atten)
    wsum = amat.sum(2)
    pred = (amat / (wsum.unsqueeze(2) + 1e-6)) @ pts2
    p, ws, w = (t.cpu() for t in ops.fine_assign(atten.cuda(), pts2.cuda()))
    assert torch.equal(w, w1)
    assert (ws - wsum).abs().max() < 1e-5 * max(1.0, wsum.abs().max().item())
    assert (p - pred).abs().max() < 2e-5


@pytest.mark.parametrize("B,M", [(2, 2049), (1, 300), (3, 65), (1, 258)])
def test_fine_match_vs_oracle(ops, B, M):
    """Similarity + assignment fused (s6d_fine_match_f32: split-bf16 MFMA tiles, three sweeps, no (B,M,M) matrix) against the
    reference chain compute_feature_similarity -> compute_fine_Rt head.  Tolerances: labels exact, weights and assigned points
    weights 2e-5 relative, assigned points 5e-5 absolute on |pts| <= 4.5 (the split drops lo.lo and the rounding of lo: <= 3 x 2^-18 of a product, times log2(e) / temp = 14.4 in the exponent; measured 1.7e-5 relative on the weights at B = 32)."""
    _check_fine_match(ops, B, M, M)


def test_fine_match_ragged_sides(ops):
    _check_fine_match(ops, 2, 131, 97)


def _check_fine_match(ops, B, M1, M2):
    g = torch.Generator().manual_seed(M1 + 7 * M2)
    f1 = torch.randn(B, M1, 256, generator=g) * (0.5 + torch.rand(B, M1, 1, generator=g))     # un-normalised out_proj rows
    perm = torch.randint(0, M1, (M2,), generator=g)
    f2 = f1[:, perm] + 0.4 * torch.randn(B, M2, 256, generator=g)
    pts2 = torch.randn(B, M2 - 1, 3, generator=g)
    atten = opem.feature_similarity(f1, f2, 0.1)
    amat, w1, _ = opem.soft_assignment(atten)
    wsum = amat.sum(2)
    pred = (amat / (wsum.unsqueeze(2) + 1e-6)) @ pts2
    p, ws, w = (t.cpu() for t in ops.fine_match(f1.cuda(), f2.cuda(), pts2.cuda(), 0.1))
    assert torch.equal(w, w1), (w != w1).sum()
    assert (ws - wsum).abs().max() < 2e-5 * max(1.0, wsum.abs().max().item()), (ws - wsum).abs().max()
    assert (p - pred).abs().max() < 5e-5, (p - pred).abs().max()


def test_positional_encoding_fused_vs_oracle(ops):
    """Fused ball-query-group + SharedMLP + max kernel (3-term split-bf16 MFMA chain since round 5; exact-f32 MFMA before) vs the
    reference PositionalEncoding."""
    from sam6d_amd.pem.pose_estimation_model import PositionalEncoding
    from sam6d_amd.utils import seeded
    pe = PositionalEncoding(256).eval()
    seeded.load_seeded(pe, 6)
    W = {"PE." + k: v for k, v in pe.state_dict().items()}
    inp = synth.pem_inputs(2, seed=9, with_rgb=False)
    pts = inp["dense_po"] / (inp["dense_po"].norm(dim=2).max(1)[0].reshape(-1, 1, 1) + 1e-6)
    with torch.no_grad():
        ref = opem.positional_encoding(W, "PE", pts)
        assert ops.have("pe_group")
        out = pe.cuda()(pts.cuda()).cpu()
    from tests.util import record_margin
    record_margin("positional_encoding_fused_vs_oracle", max_abs=(out - ref).abs().max(), bound=5e-5, ref_absmax=ref.abs().max())
    assert (out - ref).abs().max() < 5e-5, (out - ref).abs().max()


def test_cross_attention_and_linear_attention_vs_oracle(ops):
    """Fused MHA rows kernel and focused-feature-map kernel inside the product layers vs the oracle layers."""
    from sam6d_amd.pem.layers import LinearTransformerLayer, TransformerLayer
    from sam6d_amd.utils import seeded
    g = torch.Generator().manual_seed(8)
    x, mem = torch.randn(3, 197, 256, generator=g), torch.randn(3, 150, 256, generator=g)
    tl = seeded.load_seeded(TransformerLayer(256).eval(), 5)
    W = {"t." + k: v for k, v in tl.state_dict().items()}
    with torch.no_grad():
        ref = opem.cross_layer(W, "t", x, mem)
        assert ops.have("mha") and ops.have("linear_attn_focus")
        out = tl.cuda()(x.cuda(), mem.cuda()).cpu()
    assert (out - ref).abs().max() < 2e-5
    ll = seeded.load_seeded(LinearTransformerLayer(256).eval(), 6)
    W = {"l." + k: v for k, v in ll.state_dict().items()}
    xd, ms = torch.randn(2, 2048, 256, generator=g), torch.randn(2, 196, 256, generator=g)
    with torch.no_grad():
        ref = opem.linear_layer(W, "l", xd, ms)
        out = ll.cuda()(xd.cuda(), ms.cuda()).cpu()
    assert (out - ref).abs().max() < 5e-5, (out - ref).abs().max()


def test_fine_match_properties_at_the_benched_size(ops):
    """B = 32, 2049 x 2049 (BASELINE configs[1]; too large for the CPU oracle in seconds): (1) against the other device
    formulation (library bmm -> s6d_fine_assign_f32, itself oracle-checked at small sizes): identical labels, weights 5e-5
    relative; (2) equivariance: permuting the observed rows (background row kept) permutes the outputs -- every owner row is reduced
    on its own; (3) weights are probabilities mass: 0 <= wsum <= 1 + 1e-5, and rows labelled background
    carry exactly zero weight and a zero point."""
    g = torch.Generator().manual_seed(11)
    B, M = 32, 2049
    f1 = torch.randn(B, M, 256, generator=g).cuda()
    perm = torch.randperm(M, generator=g).cuda()
    f2 = f1[:, perm] + 0.4 * torch.randn(B, M, 256, generator=g).cuda()
    pts2 = torch.randn(B, M - 1, 3, generator=g).cuda()
    F = torch.nn.functional
    pred, wsum, w1 = ops.fine_match(f1, f2, pts2, 0.1)
    atten = F.normalize(f1, dim=2) @ F.normalize(f2, dim=2).transpose(1, 2) / 0.1
    p0, ws0, l0 = ops.fine_assign(atten, pts2)
    assert torch.equal(w1, l0)
    assert ((wsum - ws0).abs() <= 5e-5 * ws0.abs().clamp(min=1e-3)).all() and (pred - p0).abs().max() < 1e-4
    rp = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(M - 1, generator=g)]).cuda()
    pred2, wsum2, w12 = ops.fine_match(f1[:, rp].contiguous(), f2, pts2, 0.1)
    # labels exactly; weights and points to fp32 rounding (a row's running-maximum updates are decided per 16-row strip, so the
    # reference point of its exponentials, not their ratio, depends on its neighbours)
    assert torch.equal(w12, w1[:, rp[1:] - 1])
    assert torch.allclose(wsum2, wsum[:, rp[1:] - 1], rtol=2e-5, atol=1e-7) and torch.allclose(pred2, pred[:, rp[1:] - 1], rtol=2e-5, atol=2e-6)
    assert (wsum >= 0).all() and (wsum <= 1 + 1e-5).all()
    off = w1 == 0
    assert (wsum[off] == 0).all() and (pred[off] == 0).all()


def test_rpe_attention_rows_do_not_depend_on_the_batch_size(ops):
    """The RPE core picks its keys-per-trip instantiation by the number of query rows (four keys per trip up to 5000 rows, one beyond:
    csrc/s6d_rpe.hip); the instantiations must agree BIT FOR BIT, or a frame alone (10 instances) and the same frame inside a
    launch group (80 instances) would get different poses.  Two instances alone against the same two as rows of a batch of 28."""
    N, C, Bbig = 197, 256, 28                                  # 28 x 197 = 5516 rows: the one-key instantiation
    g = torch.Generator(device="cuda").manual_seed(12)
    q, k, v = (torch.randn(Bbig, N, C, generator=g, device="cuda") for _ in range(3))
    qt = torch.randn(Bbig, 4, N, C, generator=g, device="cuda") * 0.1
    qb = torch.randn(Bbig, 4, N, generator=g, device="cuda")
    emb = torch.randn(Bbig, N, N, C, generator=g, device="cuda")
    big = ops.rpe_attention(q, k, v, qt, qb, emb, 0.125)
    small = ops.rpe_attention(q[5:7].contiguous(), k[5:7].contiguous(), v[5:7].contiguous(), qt[5:7].contiguous(), qb[5:7].contiguous(),
                              emb[5:7].contiguous(), 0.125)
    assert torch.equal(big[5:7], small)
    ref = torch.softmax((torch.einsum("bnhc,bmhc->bhnm", q[5:7].view(2, N, 4, 64), k[5:7].view(2, N, 4, 64)) +
                         torch.einsum("bhnc,bnmc->bhnm", qt[5:7], emb[5:7]) + qb[5:7][..., None]) * 0.125, -1)
    ref = torch.einsum("bhnm,bmhc->bnhc", ref, v[5:7].view(2, N, 4, 64)).reshape(2, N, C)
    assert (small - ref).abs().max() < 2e-4 * ref.abs().max().clamp_min(1.0)
