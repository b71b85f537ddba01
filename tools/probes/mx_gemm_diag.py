"""Where the MX forms of the fp8 GEMM go wrong on the device (diagnostic)."""
import sys
import torch
sys.path.insert(0, ".")
from sam6d_amd import ops
from sam6d_amd.utils import fp8

g = torch.Generator().manual_seed(0)
for (M, N, K) in ((256, 256, 128), (256, 256, 256), (512, 256, 128)):
    # integer operands, unit weight scales: exact arithmetic
    a = torch.randint(-3, 4, (M, K), generator=g).float() * torch.exp2(torch.randint(-2, 3, (M, K // 32), generator=g).float()).repeat_interleave(32, 1)
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    qa, sa = fp8.quantize_blocks(a)
    qw, sw = fp8.quantize_rows(w)
    out = ops.gemm_fp8_mxa(qa.cuda(), sa.cuda(), qw.cuda(), sw.cuda()).float().cpu()
    ref = fp8.dequantize_blocks(qa, sa) @ fp8.dequantize_rows(qw, sw).t()
    bad = (out - ref).abs() > 1e-3 * ref.abs().clamp(min=1)
    print(f"mxa {M}x{N}x{K}: wrong {int(bad.sum())} of {bad.numel()}; rows with errors {bad.any(1).nonzero().flatten()[:12].tolist()} ... cols {bad.any(0).nonzero().flatten()[:12].tolist()}")
    if bad.any():
        i, j = bad.nonzero()[0].tolist()
        # which single-block scale assignment would explain out[i, j]?
        da = qa.view(torch.float8_e4m3fn).float()[i].view(K // 32, 32)
        dw = fp8.dequantize_rows(qw, sw)[j].view(K // 32, 32)
        parts = (da * dw).sum(1)
        print("   out", out[i, j].item(), "ref", ref[i, j].item(), "block partial sums (unscaled)", parts.tolist(), "scales", (sa[i].int() - 127).tolist())
    # uniform per-row scales through the MX path must equal the per-row kernel
    qa2, sa2 = fp8.quantize_rows(a)
    o1 = ops.gemm_fp8(qa2.cuda(), sa2.cuda(), qw.cuda(), sw.cuda()).float().cpu()
    o2 = ops.gemm_fp8_mxa(qa2.cuda(), sa2[:, None].expand(M, K // 32).contiguous().cuda(), qw.cuda(), sw.cuda()).float().cpu()
    print(f"   uniform scales: mxa == row kernel: {torch.equal(o1, o2)}  (max diff {(o1 - o2).abs().max().item()})")
    # GELU + MX output
    b = torch.zeros(N)
    q, s = ops.gemm_fp8_gelu_mx(qa2.cuda(), sa2.cuda(), qw.cuda(), sw.cuda(), b.cuda())
    refg = torch.nn.functional.gelu(fp8.dequantize_rows(qa2, sa2).double() @ fp8.dequantize_rows(qw, sw).double().t()).float()
    rq, rs = fp8.quantize_blocks(refg)
    ds = (s.cpu().int() - rs.int())
    got = fp8.dequantize_blocks(q.cpu(), s.cpu())
    amax = refg.view(M, N // 32, 32).abs().amax(2, keepdim=True).expand(-1, -1, 32).reshape(M, N)
    badv = (got - refg).abs() > 2.0 ** -4 * amax * 1.01 + 1e-30
    print(f"gelu_mx {M}x{N}x{K}: scale bytes off {int((ds != 0).sum())} of {ds.numel()} (max |d| {ds.abs().max().item()}, first rows {(ds != 0).any(1).nonzero().flatten()[:8].tolist()}); values off {int(badv.sum())} of {badv.numel()}")
    if (ds != 0).any():
        i, j = (ds != 0).nonzero()[0].tolist()
        print("   block", (i, j), "scale got", int(s[i, j]), "want", int(rs[i, j]), "ref block amax", refg[i, 32 * j:32 * j + 32].abs().max().item(), "got bytes", q[i, 32 * j:32 * j + 8].tolist(), "want", rq[i, 32 * j:32 * j + 8].tolist())
