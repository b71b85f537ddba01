"""Ablation timing of the fused attention kernels (diagnostic; run on the GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
from sam6d_amd import ops  # noqa: E402


def run(B, H, nh, hd, ws, dbg, n=5):
    # ablation masks are compile-time now: rebuild with S6D_EXTRA_HIPCC_FLAGS=-DS6D_ATTN_ABLATE=<dbg>
    g = torch.Generator().manual_seed(0)
    S = ws if ws else H
    qkv = torch.randn(B, H, H, 3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
    bias = torch.randn(3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
    rh = torch.randn(2 * S - 1, hd, generator=g).cuda().to(torch.bfloat16)
    rw = torch.randn(2 * S - 1, hd, generator=g).cuda().to(torch.bfloat16)
    for _ in range(2):
        ops.window_attention(qkv, bias, rh, rw, nh, ws, hd ** -0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.window_attention(qkv, bias, rh, rw, nh, ws, hd ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    T = S * S
    nwin = 1 if ws == 0 else ((H + ws - 1) // ws) ** 2
    flop = 4.0 * B * nwin * nh * T * T * hd
    print(f"B={B} H={H} nh={nh} ws={ws} dbg={dbg}: {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s (useful)", flush=True)


if __name__ == "__main__":
    for dbg in (0, 1, 2, 3, 1 + 4, 1 + 8, 1 + 16, 1 + 4 + 8, 1 + 4 + 16, 1 + 8 + 16, 1 + 4 + 8 + 16):
        run(8, 64, 16, 80, 0, dbg)
    for dbg in (0, 1, 2, 3):
        run(8, 64, 16, 80, 14, dbg)
    run(1, 64, 16, 80, 0, 0)
    run(1, 64, 16, 80, 14, 0)
