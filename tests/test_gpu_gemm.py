"""GPU parity of s6d_gemm_bf16 (csrc/s6d_gemm.hip: the nn.Linear layers of the ViTs with bias / exact GELU in the epilogue)
against the fp32 product of the SAME bf16-rounded operands (plain torch reference of the op: common.py:13-28,
image_encoder.py:224-240).  Tolerance: one bf16 rounding of an fp32-accumulated result (2^-8 relative); the GELU is the erf
form (torch.nn.functional.gelu, approximate='none')."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, w, bias, gelu):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    return torch.nn.functional.gelu(y) if gelu else y


def _check(out, ref, what):
    err = (out.float() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-5
    bad = int((err > 1.01 * tol).sum().item())
    assert bad == 0, f"{what}: {bad} of {ref.numel()} outside one bf16 rounding (max err {err.max().item():.3e})"


@pytest.mark.parametrize("K,N,gelu", [(1280, 3840, False), (1280, 1280, False), (1280, 5120, True), (5120, 1280, False),
                                      (1280, 256, False)])
def test_vit_h_linear_shapes_at_16_frames(K, N, gelu):
    """M = 16 frames x 4096 tokens = 65536 rows: the launch-group shapes of the benched SAM ViT-H stage."""
    from sam6d_amd import ops
    assert ops.have("gemm_bf16")
    M = 65536
    g = torch.Generator(device="cuda").manual_seed(K + N)
    a = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda")
    out = ops.gemm_bf16(a, w, b, gelu=gelu)
    for r0 in range(0, M, 8192):                       # fp32 reference in row blocks (keeps the fp32 product at 168 MB)
        _check(out[r0:r0 + 8192], _ref(a[r0:r0 + 8192], w, b, gelu), f"rows {r0}")


@pytest.mark.parametrize("M,K,N,gelu,bias,pad,blocks", [
    (197 * 32, 768, 2304, False, True, 0, 0),          # PEM ViT-B qkv at the benched batch: ragged last m-tile
    (197 * 32, 768, 3072, True, True, 0, 0),           # ... fc1 + GELU
    (257 * 7, 1024, 4096, True, True, 0, 0),           # DINOv2 ViT-L fc1, 7 crops
    (1000, 3072, 768, False, False, 0, 0),             # no bias
    (777, 192, 256, False, True, 64, 0),               # strided activation rows, one n-tile, 3 K tiles
    (255, 64, 256, True, True, 0, 0),                  # a single partial tile, one K tile
    (4096, 1280, 1280, False, True, 0, 8),             # 80 tiles on 8 workgroups: 10 output tiles per persistent workgroup
    (3000, 640, 512, True, True, 0, 16),
])
def test_shapes_edges_and_persistent_streams(M, K, N, gelu, bias, pad, blocks):
    from sam6d_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    a = torch.randn(M, K + pad, generator=g, device="cuda").to(torch.bfloat16)[:, :K]
    w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda") if bias else None
    out = ops.gemm_bf16(a, w, b, gelu=gelu, max_blocks=blocks)
    _check(out, _ref(a, w, b, gelu), "whole matrix")


def test_repeated_launches_are_bit_identical_and_rows_past_m_untouched():
    """The counted-wait pipeline has no data-dependent path: 20 launches must agree bit for bit (a schedule that reads a
    slot before its DMA has landed shows up as run-to-run differences); the output buffer behind row M is not written."""
    from sam6d_amd import ops
    M, K, N = 8192 + 100, 1280, 1280
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda")
    buf = torch.full((M + 256, N), 7.0, dtype=torch.bfloat16, device="cuda")
    first = None
    for i in range(20):
        ops.gemm_bf16(a, w, b, gelu=True, out=buf[:M])
        if first is None:
            first = buf[:M].clone()
        else:
            assert torch.equal(first, buf[:M]), f"launch {i} differs from launch 0"
    assert bool((buf[M:] == 7.0).all())
    _check(first, _ref(a, w, b, True), "whole matrix")


def test_gelu_epilogue_against_exact_gelu_over_the_whole_bf16_range():
    """x W^T with W = identity blocks exposes the epilogue alone: every finite bf16 value in [-12, 12] through the fused GELU
    equals bf16(gelu_fp32(x)) up to the rounding of a 1.5e-7-accurate erfc."""
    from sam6d_amd import ops
    vals = torch.arange(-2 ** 15, 2 ** 15, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()
    vals = vals[torch.isfinite(vals) & (vals.abs() <= 12)]
    n = vals.numel()
    M = (n + 255) // 256
    x = torch.zeros(M * 256, dtype=torch.float32)
    x[:n] = vals
    a = x.view(M, 256).to(torch.bfloat16).cuda()              # 256 test values per row, passed through W = identity
    w = torch.eye(256, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm_bf16(a, w, None, gelu=True).float().cpu().view(-1)[:n]
    ref = torch.nn.functional.gelu(vals.double()).float()
    err = (out - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 3e-7
    assert int((err > 1.01 * tol).sum()) == 0, err.max().item()


def test_sam_block_with_the_kernel_equals_the_library_statement():
    """One ViT-H block's Linear layers through fused_linear vs torch.nn.functional on the same bf16 tensors."""
    from sam6d_amd.sam.image_encoder import MLPBlock
    from sam6d_amd.utils import seeded
    mlp = seeded.load_seeded(MLPBlock(1280, 5120).eval(), 5).cuda().to(torch.bfloat16)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2, 64, 64, 1280, generator=g, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        y = mlp(x).float()
        h = torch.nn.functional.gelu(x.float() @ mlp.lin1.weight.float().t() + mlp.lin1.bias.float()).to(torch.bfloat16)
        ref = h.float() @ mlp.lin2.weight.float().t() + mlp.lin2.bias.float()
    rel = (y - ref).norm() / ref.norm()
    assert rel < 4e-3, rel.item()                                 # two bf16 roundings (h, y)


@pytest.mark.parametrize("M,N,K,cb", [(65536, 3840, 1280, 80), (700, 768, 192, 64), (300, 256, 64, 32)])
def test_column_block_output_equals_the_plain_product(M, N, K, cb):
    """s6d_gemm_bf16_cblk: column blocks stored as separate (M, col_block) matrices (the head-major q/k/v of the attention kernels) ==
    the plain (M, N) output of the same launch, rearranged -- bit for bit (only store addresses differ)."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    plain = ops.gemm_bf16(a, w, b)
    blk = ops.gemm_bf16(a, w, b, col_block=cb)
    assert blk.shape == (N // cb, M, cb)
    assert torch.equal(blk, plain.view(M, N // cb, cb).permute(1, 0, 2))


@pytest.mark.parametrize("M,N,K,inplace", [(700, 768, 192, False), (300, 256, 64, True), (65536, 1280, 1280, True), (4096, 1280, 5120, False)])
def test_residual_gemm_sums_in_the_accumulators(M, N, K, inplace):
    """s6d_gemm_bf16_res: x + Linear(a) with the accumulators started at bias + residual: the fp32 sum a W^T + b + x rounded to bf16
    ONCE (one bf16 rounding of the float reference), also in place and with ragged M; zero activations return the bias + residual
    exactly."""
    from sam6d_amd import ops
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    x = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    ref = _ref(a, w, b, False) + x.float()
    exact0 = (b[None, :] + x.float()).to(torch.bfloat16)
    zero = ops.gemm_bf16(torch.zeros_like(a), w, b, residual=x)
    assert torch.equal(zero, exact0)
    got = ops.gemm_bf16(a, w, b, residual=x, out=x if inplace else None)
    assert (got.data_ptr() == x.data_ptr()) == inplace
    _check(got, ref, "residual sum")


@pytest.mark.parametrize("M,N,K", [(700, 768, 192), (300, 256, 64), (260, 1280, 64), (300, 1024, 128), (65536, 1280, 1280)])
def test_residual_gemm_row_statistics(M, N, K):
    """The partial LayerNorm statistics the residual GEMM emits, combined by s6d_ln_stats_finalize, are the mean / sigma = sqrt(var + eps) of the fp32
    result rows (float64 reference), and s6d_row_stats_bf16 of the stored bf16 rows agrees with them to the bf16 rounding."""
    from sam6d_amd import ops
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(N, generator=g) + 0.5).cuda()
    x = (torch.randn(M, N, generator=g) * torch.rand(M, 1, generator=g) * 3 + torch.randn(M, 1, generator=g)).to(torch.bfloat16).cuda()
    sp = torch.full((N // 32, 2, M), float("nan"), device=a.device)
    got = ops.gemm_bf16(a, w, b, residual=x, stats_partial=sp)
    st = ops.ln_stats_finalize(sp, 32, 1e-6)
    ref = (a.double() @ w.double().t() + b.double() + x.double())
    mean = ref.mean(1)
    sigma = torch.sqrt(ref.var(1, unbiased=False) + 1e-6)
    assert torch.isfinite(st).all()
    assert (st[:, 0].double() - mean).abs().max().item() <= 2e-6 * (1 + ref.abs().max().item())
    assert ((st[:, 1].double() - sigma).abs() / sigma).max().item() <= 2e-5
    st2 = ops.row_stats(got, 1e-6)
    assert (st2[:, 0] - st[:, 0]).abs().max().item() <= 1e-3 * (1 + ref.abs().max().item()) / N ** 0.5 * 4
    assert ((st2[:, 1] - st[:, 1]).abs() / st[:, 1]).max().item() <= 2e-3


@pytest.mark.parametrize("M,N,K,gelu,cb", [(700, 768, 192, False, 0), (300, 256, 320, True, 0), (520, 768, 256, False, 64),
                                             (65536, 3840, 1280, False, 0), (16384, 5120, 1280, True, 0)])
def test_lnfold_gemm_vs_layernorm_then_linear(M, N, K, gelu, cb):
    """s6d_gemm_bf16_lnfold: act(LN(x) W^T + b) from the RAW rows, their (mean, sigma) and the folded weight.  Against the float64
    evaluation of the same folded form: one bf16 rounding.  Against LayerNorm-then-Linear in float64 (the statement it replaces,
    segment_anything/modeling/image_encoder.py:166-182): the rounding of gamma * W to bf16, relative rms below 3e-3."""
    from sam6d_amd import ops
    from sam6d_amd.utils.linear import lnfold_weights
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(M + N + K + 2)
    x = (torch.randn(M, K, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g)).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).cuda()
    beta = (0.2 * torch.randn(K, generator=g)).cuda()
    wf, cs, bf = lnfold_weights(W, b, gamma, beta)
    st = ops.row_stats(x, 1e-6)
    out = ops.gemm_bf16_lnfold(x, st, wf, cs, bf, gelu=gelu, col_block=cb)
    if cb:
        assert out.shape == (N // cb, M, cb)
        out = out.permute(1, 0, 2).reshape(M, N)
    xd = x.double()
    mu, rs = st[:, :1].double(), 1.0 / st[:, 1:].double()
    folded = rs * (xd @ wf.double().t() - mu * cs.double()[None, :]) + bf.double()[None, :]
    true = torch.nn.functional.layer_norm(xd, (K,), gamma.double(), beta.double(), 1e-6) @ W.double().t() + b.double()
    if gelu:
        folded, true = torch.nn.functional.gelu(folded), torch.nn.functional.gelu(true)
    _check(out, folded.float(), "folded form")
    rel = ((out.double() - true).pow(2).mean() / true.pow(2).mean()).sqrt().item()
    assert rel <= 3e-3, rel


@pytest.mark.parametrize("M,N,K,gelu", [(300, 256, 320, True), (700, 768, 192, False), (6304, 3072, 768, True), (6304, 768, 3072, False)])
def test_float16_gemm_vs_float(M, N, K, gelu):
    """s6d_gemm_f16 (the bf16 kernel on v_mfma_f32_32x32x16_f16, half pack): fp32 accumulation, one rounding to half."""
    from sam6d_amd import ops
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.float16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.float16).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = ops.gemm_bf16(a, w, b, gelu=gelu)
    assert out.dtype == torch.float16
    ref = a.float().cpu().double() @ w.float().cpu().double().t() + b.cpu().double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = (out.float().cpu().double() - ref).abs()
    assert (err <= 2.0 ** -11 * ref.abs() * 1.01 + 2e-5).all(), (err / (2.0 ** -11 * ref.abs() + 2e-5)).max().item()


@pytest.mark.parametrize("M,N,K", [(300, 256, 320), (8192, 3840, 1280)])
def test_lnfold_gemm_with_offset_rows(M, N, K):
    """The fold subtracts mean_m s_n from a product of the RAW rows: rows whose mean is 30 standard deviations (bf16 spacing at that
    magnitude: 1/8 of sigma -- the stream's own rounding, which the LayerNorm pass of the unfolded form reads too) must come out as
    close to the float64 LayerNorm -> Linear of the same bf16 rows as rows with zero mean do."""
    from sam6d_amd import ops
    from sam6d_amd.utils.linear import lnfold_weights
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(M + N + K + 3)
    base = torch.randn(M, K, generator=g)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    gamma = (1 + 0.3 * torch.randn(K, generator=g)).cuda()
    beta = (0.2 * torch.randn(K, generator=g)).cuda()
    wf, cs, bf = lnfold_weights(W, b, gamma, beta)
    rel = {}
    for off in (0.0, 30.0):
        x = (base + off).to(torch.bfloat16).cuda()
        out = ops.gemm_bf16_lnfold(x, ops.row_stats(x, 1e-6), wf, cs, bf)
        true = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-6) @ W.double().t() + b.double()
        rel[off] = ((out.double() - true).pow(2).mean() / true.pow(2).mean()).sqrt().item()
    assert rel[0.0] <= 3e-3 and rel[30.0] <= 1.5 * rel[0.0] + 1e-3, rel


@pytest.mark.parametrize("M,N,K,kind,blocks", [
    (65536, 3840, 1280, "lnfold_cblk", 0),     # qkv of 16 ViT-H frames: folded LayerNorm, head-major column blocks
    (16384, 5120, 1280, "lnfold_gelu", 0),     # lin1 + GELU
    (4096, 1280, 5120, "plain", 0),            # 80 K tiles per output tile
    (256, 256, 128, "gelu", 0),                # the smallest shape of the form: one tile, two K tiles
    (2560, 512, 128, "plain", 8),              # 20 tiles on 8 workgroups, two K tiles each: the stream changes tile in every iteration
    (1536, 768, 192, "gelu", 8),               # 18 tiles on 8 workgroups, three K tiles (the ring wraps inside a tile)
    (1024, 768, 768, "f16_gelu", 0),           # IEEE-half operands
    (768, 256, 1280, "nobias", 8),
    (65536, 1280, 1280, "res_stats", 0),       # proj of 16 ViT-H frames: residual in the accumulators + partial row statistics, in place
    (4096, 1280, 5120, "res_stats", 0),        # lin2
    (2560, 512, 128, "res", 8),                # residual prefetch across tiles: 20 tiles on 8 workgroups, two K tiles each
    (1536, 768, 192, "res_nobias", 8),
])
def test_four_wave_form_gives_the_bits_of_the_eight_wave_form(M, N, K, kind, blocks):
    """csrc/s6d_gemm4.hip (four waves, 128 x 128 wave tiles, K loop in assembly, accumulators in the accumulator register file) against
    csrc/s6d_gemm.hip through s6d_set_gemm_wave_tile: the same products in the same order per accumulator and the same epilogue
    arithmetic -> equal bits, on every epilogue the form covers; ten repeated launches agree bit for bit (a fragment read before its
    DMA piece landed would show as run-to-run differences); the result is one rounding from the float product."""
    from sam6d_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dt = torch.float16 if kind.startswith("f16") else torch.bfloat16
    a = (torch.randn(M, K, generator=g, device="cuda") * (0.5 + torch.rand(M, 1, generator=g, device="cuda"))).to(dt)
    w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(dt)
    b = None if kind == "nobias" else torch.randn(N, generator=g, device="cuda")
    if kind.startswith("lnfold"):
        st = ops.row_stats(a, 1e-6)
        cs = w.float().sum(1).contiguous()
        fn = lambda: ops.gemm_bf16_lnfold(a, st, w, cs, b, gelu=kind == "lnfold_gelu", col_block=80 if kind == "lnfold_cblk" else 0,   # noqa: E731
                                          max_blocks=blocks)
    elif kind.startswith("res"):
        if kind == "res_nobias":
            b = None
        r = torch.randn(M, N, generator=g, device="cuda").to(dt)
        sp = torch.full((N // 32, 2, M), float("nan"), device="cuda") if kind == "res_stats" else None
        keep = {}

        def fn():
            x = r.clone()                                   # in place on the residual (what the ViT blocks do)
            out = ops.gemm_bf16(a, w, b, residual=x, out=x, stats_partial=sp, max_blocks=blocks)
            if sp is not None:
                keep["sp"] = sp.clone()
            return out
    else:
        fn = lambda: ops.gemm_bf16(a, w, b, gelu=kind.endswith("gelu"), max_blocks=blocks)   # noqa: E731
    try:
        ops.set_gemm_wave_tile(64)
        ref = fn().clone()
        sp_ref = keep["sp"] if kind == "res_stats" else None
        ops.set_gemm_wave_tile(128)
        got = fn().clone()
        assert torch.equal(got, ref), f"{int((got != ref).sum())} of {ref.numel()} values differ from the eight-wave form"
        if sp_ref is not None:                              # the partial LayerNorm statistics too, bit for bit
            assert torch.equal(keep["sp"], sp_ref) and torch.isfinite(sp_ref).all()
        for i in range(10):
            assert torch.equal(fn(), got), f"launch {i} differs"
    finally:
        ops.set_gemm_wave_tile(0)
    if not kind.startswith("lnfold"):
        want = _ref(a, w, b, kind.endswith("gelu")) + (r.float() if kind.startswith("res") else 0.0)
        err = (got.float() - want).abs()
        tol = (2.0 ** -11 if dt == torch.float16 else 2.0 ** -8) * want.abs() + 1e-5
        assert int((err > 1.01 * tol).sum()) == 0, err.max().item()


def test_wave_tile_switch_rejects_other_values():
    from sam6d_amd import _lib
    L = _lib.lib()
    assert L.s6d_set_gemm_wave_tile(32) == -1
    assert L.s6d_set_gemm_wave_tile(0) == 0


@pytest.mark.parametrize("M,N,K,gelu,half,blocks", [(6304, 768, 3072, False, True, 0), (6304, 768, 768, False, True, 0), (1970, 2304, 768, False, True, 0),
                                                   (1970, 3072, 768, True, True, 0), (4096, 256, 1280, False, False, 0), (300, 256, 320, True, False, 0),
                                                   (700, 768, 192, False, False, 16), (261, 512, 64, True, True, 8)])
def test_small_tile_form_gives_the_bits_of_the_256_tile_form(M, N, K, gelu, half, blocks):
    """Round 6: plain / GELU launches that would put fewer than 160 tiles of 256 x 256 on the chip take the 256 x 128 kernel
    (gemm2_bf16_kernel<., ., DT>, now also in IEEE half: the PEM ViT-B's products).  Same products in the same order per element:
    the result must equal the 256 x 256 kernel's bit for bit (s6d_set_gemm_small_tile(0)), ragged row counts and few persistent
    workgroups included -- the PEM's group = single property rests on it (the instantiation follows the row count)."""
    from sam6d_amd import ops
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    dt = torch.float16 if half else torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    try:
        ops.set_gemm_small_tile(False)
        big = ops.gemm_bf16(a, w, b, gelu=gelu, max_blocks=blocks)
        ops.set_gemm_small_tile(True)
        small = ops.gemm_bf16(a, w, b, gelu=gelu, max_blocks=blocks)
    finally:
        ops.set_gemm_small_tile(True)
    assert small.dtype == dt and torch.equal(small, big)
    ref = a.float().cpu().double() @ w.float().cpu().double().t() + b.cpu().double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    eps = 2.0 ** -11 if half else 2.0 ** -8
    err = (small.float().cpu().double() - ref).abs()
    assert (err <= eps * ref.abs() * 1.01 + 1e-3).all()


@pytest.mark.parametrize("M,N,K,inplace,stats,bias", [(4096, 1280, 1280, True, True, True), (4096, 1280, 5120, False, True, True),
                                                     (700, 768, 192, False, True, True), (300, 256, 64, True, False, True),
                                                     (261, 1280, 128, False, True, False)])
def test_small_tile_residual_form_gives_the_bits_of_the_256_tile_form(M, N, K, inplace, stats, bias):
    """Round 6: the residual + row-statistics epilogue (EPI 2: the ViT-H's proj / lin2) in the 256 x 128 kernel for launches whose 256 x 256
    tiling would put fewer than 160 tiles on the chip (one frame: 4096 x 1280 = 80 tiles) -- selectable (s6d_set_gemm_small_tile(2)), not
    the default: the encoder on one frame measured 10.73 ms with it against 10.11 ms without.  Its lanes hold other columns than the
    eight-wave kernel's; one v_permlane32_swap per register pair restores that kernel's 32 consecutive columns per lane, so the
    outputs AND the partial statistics (summed in the same ascending order) must be bit-identical -- with ragged rows, in place,
    without statistics and without a bias."""
    from sam6d_amd import ops
    if not torch.cuda.is_available() and M > 1000:
        pytest.skip("emulator: small shapes only")
    g = torch.Generator().manual_seed(M + N + K + 7)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = (torch.randn(N, generator=g) + 0.5).cuda() if bias else None
    x = (torch.randn(M, N, generator=g) * 2 + torch.randn(M, 1, generator=g)).to(torch.bfloat16).cuda()
    res = {}
    try:
        for small in (False, True):
            ops.set_gemm_small_tile(2 if small else 0)                 # 2: the residual epilogue too (not in the default: it is slower)
            xr = x.clone()
            sp = torch.full((N // 32, 2, M), float("nan"), device=a.device) if stats else None
            out = ops.gemm_bf16(a, w, b, residual=xr, out=xr if inplace else None, stats_partial=sp)
            res[small] = (out.clone(), None if sp is None else sp.clone())
    finally:
        ops.set_gemm_small_tile(True)
    assert torch.equal(res[True][0], res[False][0])
    if stats:
        assert torch.isfinite(res[True][1]).all() and torch.equal(res[True][1], res[False][1])
    ref = a.float() @ w.float().t() + (b if bias else 0) + x.float()
    _check(res[True][0], ref, "residual sum, 256 x 128 tiles")
