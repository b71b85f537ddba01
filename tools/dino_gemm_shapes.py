"""The four GEMMs of a DINOv2 ViT-L block at the benched crop count, in the forms the block loop uses (LayerNorm-folded qkv and
fc1 + GELU, residual + statistics proj and fc2), against the same launches at a row count that is a whole number of 256-row tiles:
what the half-filled last row tile of 128 crops x 257 tokens = 32896 = 128.5 tiles costs.  usage: dino_gemm_shapes.py [crops]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam6d_amd import ops  # noqa: E402


def ev(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    crops = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    g = torch.Generator(device="cuda").manual_seed(0)
    C = 1024
    rows = {}
    for M in (crops * 257, (crops * 257) // 256 * 256, ((crops * 257) + 255) // 256 * 256):
        x = torch.randn(M, C, generator=g, device="cuda").bfloat16()
        h = torch.randn(M, 4 * C, generator=g, device="cuda").bfloat16()
        st = ops.row_stats(x, 1e-6)
        for name, K, N, kind in (("qkv", C, 3 * C, "fold"), ("proj", C, C, "res"), ("fc1+gelu", C, 4 * C, "foldg"), ("fc2", 4 * C, C, "res")):
            w = (0.02 * torch.randn(N, K, generator=g, device="cuda")).bfloat16()
            b = torch.zeros(N, device="cuda")
            cs = w.float().sum(1)
            a = x if K == C else h
            sp = torch.empty(N // 32, 2, M, dtype=torch.float32, device="cuda")
            r = torch.randn(M, N, generator=g, device="cuda").bfloat16()
            if kind == "res":
                fn = lambda: ops.gemm_bf16(a, w, b, residual=r, out=r, stats_partial=sp)
            else:
                fn = lambda: ops.gemm_bf16_lnfold(a, st, w, cs, b, gelu=(kind == "foldg"))
            ms = min(ev(fn) for _ in range(3))
            rows[f"M={M} {name}"] = {"us": round(ms * 1e3, 1), "tflops": round(2.0 * M * N * K / ms / 1e9, 1),
                                     "tiles": ((M + 255) // 256) * (N // 256)}
            print(f"M={M:6d} {name:9s} {json.dumps(rows[f'M={M} {name}'])}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/dino_gemm_shapes.json", "w"), indent=1)


if __name__ == "__main__":
    main()
