"""Device diagnostics of s6d_gemm_fp8: exact cases (small integers, power-of-two row scales) and the error of random operands
relative to sum |a||w| -- separates an operand-layout / scale-routing bug from the matrix instruction's internal precision."""
import sys

import torch

sys.path.insert(0, ".")
from sam6d_amd import ops  # noqa: E402
from sam6d_amd.utils import fp8  # noqa: E402

g = torch.Generator().manual_seed(0)
for M, N, K in ((256, 256, 128), (512, 512, 512), (300, 768, 1280)):
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    a[:, ::7] = 0
    ea = torch.randint(-10, 11, (M,), generator=g)
    ew = torch.randint(-5, 6, (N,), generator=g)
    qa, qw = a.to(torch.float8_e4m3fn).view(torch.uint8), w.to(torch.float8_e4m3fn).view(torch.uint8)
    sa, sw = (ea + 127).to(torch.uint8), (ew + 127).to(torch.uint8)
    out = ops.gemm_fp8(qa.cuda(), sa.cuda(), qw.cuda(), sw.cuda()).float().cpu()
    ref = torch.ldexp(a, ea[:, None]).double() @ torch.ldexp(w, ew[:, None]).double().t()
    refb = ref.float().to(torch.bfloat16).float()
    print(f"int M={M} N={N} K={K}: exact {torch.equal(out, refb)}  mismatches {(out != refb).sum().item()} of {out.numel()}  max |rel| {((out - refb).abs() / refb.abs().clamp(min=1e-30)).max().item():.3e}")
    if not torch.equal(out, refb):
        bad = (out != refb).nonzero()[:5]
        for i, j in bad.tolist():
            print("   ", i, j, out[i, j].item(), refb[i, j].item())
for M, N, K in ((256, 256, 128), (512, 256, 1280)):
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    qa, sa = fp8.quantize_rows(a)
    qw, sw = fp8.quantize_rows(w)
    out = ops.gemm_fp8(qa.cuda(), sa.cuda(), qw.cuda(), sw.cuda()).float().cpu()
    da, dw = fp8.dequantize_rows(qa, sa).double(), fp8.dequantize_rows(qw, sw).double()
    ref = da @ dw.t()
    scale = da.abs() @ dw.abs().t()
    err = (out.double() - ref).abs()
    print(f"rand M={M} N={N} K={K}: max err/scale {(err / scale).max().item():.3e}  max err/|ref| {(err / ref.abs()).max().item():.3e}  median err/|ref| {(err / ref.abs()).median().item():.3e}  (bf16 half-ulp 3.9e-3)")
    # K-block structure: error of a product over ONE 64-wide block (the instruction itself) and over 128
    for kk in (64, 128):
        out2 = ops.gemm_fp8(qa[:, :128].contiguous().cuda(), sa.cuda(), qw[:, :128].contiguous().cuda(), sw.cuda()).float().cpu() if kk == 128 else None
