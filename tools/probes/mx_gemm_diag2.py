"""Decode which (row, block) scale byte the MX GEMM applies where (diagnostic): all-ones operands, structured scale bytes."""
import sys
import torch
sys.path.insert(0, ".")
from sam6d_amd import ops

M, N = 256, 256
for K in (128, 256):
    nb = K // 32
    qa = torch.full((M, K), 0x38, dtype=torch.uint8).cuda()            # e4m3 1.0
    qw = torch.full((N, K), 0x38, dtype=torch.uint8).cuda()
    sw = torch.full((N,), 127, dtype=torch.uint8).cuda()
    # (1) block routing: exponent = block index -> every output must be 32 * sum_b 2^b
    s = (127 + torch.arange(nb)).to(torch.uint8)[None].expand(M, nb).contiguous().cuda()
    out = ops.gemm_fp8_mxa(qa, s, qw, sw).float().cpu()
    want = 32.0 * sum(2.0 ** b for b in range(nb))
    print(f"K={K} block test: want {want}; got unique {out.unique().tolist()[:8]}")
    # (2) one block hot: exponent 4 at block b, 0 elsewhere -> 32 * (nb - 1 + 16)
    for b in range(nb):
        s = torch.full((M, nb), 127, dtype=torch.uint8)
        s[:, b] = 131
        out = ops.gemm_fp8_mxa(qa, s.cuda(), qw, sw).float().cpu()
        print(f"   hot block {b}: want {32.0 * (nb - 1 + 16)}; got unique {out.unique().tolist()[:6]}")
    # (3) row routing: exponent = row % 8 in every block -> out[i][:] = 32 nb 2^(i % 8)
    s = (127 + (torch.arange(M) % 8)).to(torch.uint8)[:, None].expand(M, nb).contiguous().cuda()
    out = ops.gemm_fp8_mxa(qa, s, qw, sw).float().cpu()
    got = torch.log2(out[:, 0] / (32.0 * nb)).round().int().tolist()
    print(f"   row test: exponent seen by output rows 0..39: {got[:40]}")
    print(f"   row test: rows 128..167: {got[128:168]}")

# (4) random bytes per (row, block), all-ones data: out[i][:] = 32 sum_b 2^(s[i][b] - 127), exact
from sam6d_amd.utils import fp8
g = torch.Generator().manual_seed(1)
for K in (128, 256, 1280):
    nb = K // 32
    qa = torch.full((M, K), 0x38, dtype=torch.uint8).cuda()
    qw = torch.full((N, K), 0x38, dtype=torch.uint8).cuda()
    sw = torch.full((N,), 127, dtype=torch.uint8).cuda()
    s = torch.randint(119, 132, (M, nb), generator=g).to(torch.uint8)
    out = ops.gemm_fp8_mxa(qa, s.cuda(), qw, sw).float().cpu()
    want = 32.0 * torch.exp2(s.float() - 127).sum(1)
    bad = (out[:, 0] - want).abs() > 1e-3 * want
    print(f"random bytes, ones data, K={K}: rows wrong {int(bad.sum())} of {M}; first {bad.nonzero().flatten()[:10].tolist()}")
    if bad.any():
        i = int(bad.nonzero()[0])
        print("   row", i, "bytes", (s[i].int() - 127).tolist(), "want", want[i].item(), "got", out[i, 0].item())
# (5) random DATA, scales uniform per row but different between rows
a = torch.randint(-3, 4, (M, 128), generator=g).float()
w = torch.randint(-2, 3, (N, 128), generator=g).float()
qa, sa = fp8.quantize_rows(a)
qw, sw = fp8.quantize_rows(w)
out = ops.gemm_fp8_mxa(qa.cuda(), sa[:, None].expand(M, 4).contiguous().cuda(), qw.cuda(), sw.cuda()).float().cpu()
ref = fp8.dequantize_rows(qa, sa) @ fp8.dequantize_rows(qw, sw).t()
print("random data, row scales via MX path: max err", (out - ref).abs().max().item())
# (6) random data, ONE block of every row rescaled by 2^-2 (data x4 in that block, scale byte - 2): same product
for hot in range(4):
    a2 = a.clone()
    qa2 = qa.clone().view(M, 4, 32)
    s2 = sa[:, None].expand(M, 4).clone()
    # requantise block `hot` with its own scale
    blk = a[:, 32 * hot:32 * hot + 32] * 0.25
    qb, sb = fp8.quantize_rows(blk)
    qa2[:, hot] = qb
    s2[:, hot] = sb
    a2[:, 32 * hot:32 * hot + 32] = blk
    out = ops.gemm_fp8_mxa(qa2.view(M, 128).contiguous().cuda(), s2.contiguous().cuda(), qw.cuda(), sw.cuda()).float().cpu()
    ref2 = a2 @ fp8.dequantize_rows(qw, sw).t()
    print(f"random data, block {hot} requantised: max err {(out - ref2).abs().max().item()}  (ref max {ref2.abs().max().item()})")
