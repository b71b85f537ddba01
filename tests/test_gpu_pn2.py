"""GPU parity: hand-written gfx950 PointNet++ ops (through the C ABI) vs the C oracle.
Bit-exact for every index output."""
import pytest
import torch

from oracle import pn2 as opn2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from sam6d_amd import ops
    return ops


def _cloud(B, N, seed, dup=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, 3, generator=g)
    if dup:  # exact duplicates and coincident points force the tie-break rule
        x[:, N // 2:] = x[:, : N - N // 2]
        x[:, 5] = x[:, 3]
    return x


@pytest.mark.parametrize("B,N,M,dup", [(2, 2048, 196, False), (3, 2048, 196, True), (1, 700, 64, True),
                                       (2, 64, 64, False), (1, 1, 1, False), (2, 3000, 128, True),
                                       (1, 4096, 200, False), (1, 20000, 300, True)])
def test_fps_bit_exact(ops, B, N, M, dup):
    x = _cloud(B, N, 100 + N, dup)
    ref = opn2.furthest_point_sampling(x, M)
    out = ops.furthest_point_sampling(x.cuda(), M)
    assert out.dtype == torch.int32 and tuple(out.shape) == (B, M)
    assert torch.equal(out.cpu(), ref)


def test_fps_tie_break_rule(ops):
    y = torch.zeros(1, 600, 3)
    y[0, 520] = 1.0
    y[0, 9] = 1.0
    assert ops.furthest_point_sampling(y.cuda(), 2).cpu()[0, 1].item() == 520
    w = torch.zeros(1, 700, 3)
    w[0, 180] = 1.0
    w[0, 530] = 1.0
    assert ops.furthest_point_sampling(w.cuda(), 2).cpu()[0, 1].item() == 180  # bit-reversed slot order
    z = torch.zeros(2, 2048, 3)
    assert torch.equal(ops.furthest_point_sampling(z.cuda(), 7).cpu(), torch.zeros(2, 7, dtype=torch.int32))


@pytest.mark.parametrize("B,N,M,r,ns", [(2, 2048, 2048, 0.1, 32), (2, 2048, 2048, 0.2, 64), (1, 300, 50, 0.2, 16),
                                        (1, 100, 7, 0.5, 128), (1, 5000, 100, 0.05, 8)])
def test_ball_query_bit_exact(ops, B, N, M, r, ns):
    x = _cloud(B, N, 7 + N, dup=True)
    q = x[:, :M].contiguous() if M <= N else _cloud(B, M, 3)
    ref = opn2.ball_query(q, x, r, ns)
    out = ops.ball_query(q.cuda(), x.cuda(), r, ns)
    assert torch.equal(out.cpu(), ref)


def test_ball_query_no_neighbour_gives_zeros(ops):
    x = _cloud(1, 256, 1)
    far = torch.full((1, 4, 3), 50.0)
    assert torch.equal(ops.ball_query(far.cuda(), x.cuda(), 0.2, 8).cpu(), torch.zeros(1, 4, 8, dtype=torch.int32))


def test_gather_group_exact(ops):
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(3, 37, 500, generator=g)
    idx = torch.randint(0, 500, (3, 196), generator=g, dtype=torch.int32)
    assert torch.equal(ops.gather_points(feats.cuda(), idx.cuda()).cpu(), opn2.gather_points(feats, idx))
    gi = torch.randint(0, 500, (3, 60, 16), generator=g, dtype=torch.int32)
    assert torch.equal(ops.group_points(feats.cuda(), gi.cuda()).cpu(), opn2.group_points(feats, gi))
    rows = torch.randn(3, 500, 256, generator=g)
    exp = torch.gather(rows, 1, idx.long().unsqueeze(-1).expand(-1, -1, 256))
    assert torch.equal(ops.gather_rows(rows.cuda(), idx.cuda()).cpu(), exp)
    rows3 = torch.randn(3, 500, 3, generator=g)
    exp3 = torch.gather(rows3, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(ops.gather_rows(rows3.cuda(), idx.cuda()).cpu(), exp3)


def test_errors_are_python_exceptions(ops):
    x = _cloud(1, 64, 1)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.furthest_point_sampling(x, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.furthest_point_sampling(x.cuda().transpose(1, 2).transpose(1, 2)[:, ::2], 4)
    with pytest.raises(RuntimeError, match="int tensor"):
        ops.gather_points(x.cuda().transpose(1, 2).contiguous(), torch.zeros(1, 4, dtype=torch.int64).cuda())
    from sam6d_amd._lib import S6DError
    with pytest.raises(S6DError):
        ops.furthest_point_sampling(x.cuda(), 1000)  # M > N


def test_fps_greedy_property_at_the_benched_batch(ops):
    """Large-size property check at BASELINE sizes (B=32): FPS output is a set of distinct
    indices, starts at 0 and greedily maximises the min-distance (checked with torch)."""
    x = _cloud(32, 2048, 11).cuda()
    idx = ops.furthest_point_sampling(x, 196).long()
    assert (idx[:, 0] == 0).all()
    assert all(len(set(r.tolist())) == 196 for r in idx.cpu())
    sel = torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    d = torch.cdist(x, sel[:, :-1])            # distance of every point to the first 195 picks
    mind = d.min(2)[0]
    last = torch.gather(mind, 1, idx[:, -1:])
    assert torch.allclose(last.squeeze(1), mind.max(1)[0], rtol=1e-5)
