"""Per-launch cost of the folded residual + LayerNorm pieces against the launches they replace, ViT-H shapes at 16 frames
(M = 65536), HIP events, two interleaved rounds of 20 launches.  -> profiles/r03_lnfold_micro.txt"""
import sys

import torch

sys.path.insert(0, ".")
from sam6d_amd import ops  # noqa: E402
from sam6d_amd.utils.linear import lnfold_weights  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, C, H = 65536, 1280, 5120


def ev(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def mk(n, k):
    return (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)


x = torch.randn(M, C, generator=g).to(dev).to(torch.bfloat16)
a = torch.randn(M, C, generator=g).to(dev).to(torch.bfloat16)
h = torch.randn(M, H, generator=g).to(dev).to(torch.bfloat16)
gam, bet = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
Wqkv, Wp, W1, W2 = mk(3 * C, C), mk(C, C), mk(H, C), mk(C, H)
bq, bp, b1, b2 = (torch.randn(n, generator=g).to(dev) for n in (3 * C, C, H, C))
wq, wp, w1, w2 = (t.to(torch.bfloat16) for t in (Wqkv, Wp, W1, W2))
fq, f1 = lnfold_weights(Wqkv, bq, gam, bet), lnfold_weights(W1, b1, gam, bet)
sp = torch.empty(C // 32, 2, M, device=dev)
st = ops.row_stats(x)
xr = x.clone()
cases = [
    ("qkv plain", lambda: ops.gemm_bf16(x, wq, bq)),
    ("qkv LN-folded", lambda: ops.gemm_bf16_lnfold(x, st, fq[0], fq[1], fq[2])),
    ("proj plain", lambda: ops.gemm_bf16(a, wp, bp)),
    ("proj + residual", lambda: ops.gemm_bf16(a, wp, bp, residual=xr, out=xr)),
    ("proj + residual + stats", lambda: ops.gemm_bf16(a, wp, bp, residual=xr, out=xr, stats_partial=sp)),
    ("lin1+gelu plain", lambda: ops.gemm_bf16(x, w1, b1, gelu=True)),
    ("lin1+gelu LN-folded", lambda: ops.gemm_bf16_lnfold(x, st, f1[0], f1[1], f1[2], gelu=True)),
    ("lin2 plain", lambda: ops.gemm_bf16(h, w2, b2)),
    ("lin2 + residual", lambda: ops.gemm_bf16(h, w2, b2, residual=xr, out=xr)),
    ("lin2 + residual + stats", lambda: ops.gemm_bf16(h, w2, b2, residual=xr, out=xr, stats_partial=sp)),
    ("add_layernorm (2 reads, 2 writes)", lambda: ops.add_layernorm(x, a, gam, bet, 1e-6)),
    ("ln_stats_finalize", lambda: ops.ln_stats_finalize(sp, 32, 1e-6)),
    ("row_stats", lambda: ops.row_stats(x)),
]
res = {}
for rnd in range(2):
    for nm, fn in cases:
        res.setdefault(nm, []).append(ev(fn))
print(f"# M = {M}; ms per launch, two interleaved rounds")
for nm, v in res.items():
    print(f"{nm:36s} {v[0]:.4f} {v[1]:.4f}")
m = {k: min(v) for k, v in res.items()}
old = m["qkv plain"] + m["proj plain"] + m["lin1+gelu plain"] + m["lin2 plain"] + 2 * m["add_layernorm (2 reads, 2 writes)"]
new = (m["qkv LN-folded"] + m["proj + residual + stats"] + m["lin1+gelu LN-folded"] + m["lin2 + residual + stats"] + 2 * m["ln_stats_finalize"])
print(f"per block (GEMMs + LayerNorm side, back to back): round-2 form {old:.4f} ms, folded form {new:.4f} ms ({100 * (new / old - 1):+.1f} %)")
