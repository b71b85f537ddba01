"""Tile-order knob of the bf16 GEMM (S6D_GEMM_GM = m tiles per n-tile group of the XCD-contiguous tile order; default 8) at the four
ViT-H shapes, back to back (the traffic side of VERDICT r3 item 4: does another panel order move time?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam6d_amd import ops  # noqa: E402
from sam6d_amd.utils.linear import lnfold_weights  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
M = 65536
res = {}
for nm, K, N, gelu, form in (("qkv", 1280, 3840, False, "lnfold"), ("proj", 1280, 1280, False, "res"), ("lin1", 1280, 5120, True, "lnfold"),
                             ("lin2", 5120, 1280, False, "res")):
    x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    if form == "lnfold":
        wf, cs, bf = lnfold_weights(W, b, torch.ones(K, device=dev), torch.zeros(K, device=dev))
        st = ops.row_stats(x)
        fn = lambda: ops.gemm_bf16_lnfold(x, st, wf, cs, bf, gelu=gelu)           # noqa: E731
    else:
        w = W.to(torch.bfloat16)
        xr = torch.randn(M, N, generator=g).to(dev).to(torch.bfloat16)
        sp = torch.empty(N // 32, 2, M, device=dev)
        fn = lambda: ops.gemm_bf16(x, w, b, residual=xr, out=xr, stats_partial=sp)  # noqa: E731
    for gm in (1, 2, 4, 8, 16, 32, 64, 256):
        os.environ["S6D_GEMM_GM"] = str(gm)
        ms = bench._event_ms(fn, 10)
        res.setdefault(nm, {})[gm] = round(ms, 4)
    print(nm, res[nm], flush=True)
