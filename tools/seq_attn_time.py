"""Sequence attention (s6d_seq_attention_bf16 / _f16) at the DINOv2 ViT-L (crops x 16 heads x 257 tokens) and PEM ViT-B (32 x 12 x 197)
shapes: HIP-event time per launch, matrix and HBM fractions.  S6D_SEQ_ATTN_IMPL=1 = the all-resident window kernel of rounds 1-3
(the choice is read once per process: run twice for an A/B).   usage: seq_attn_time.py [crops]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam6d_amd import ops  # noqa: E402


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


if __name__ == "__main__":
    crops = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    g = torch.Generator().manual_seed(0)
    for name, B, N, nh, dt in (("DINOv2 ViT-L/14", crops, 257, 16, torch.bfloat16), ("DINOv2 ViT-L/14 (255 crops)", 255, 257, 16, torch.bfloat16),
                               ("PEM ViT-B (f16)", 32, 197, 12, torch.float16), ("PEM ViT-B K=10 (f16)", 10, 197, 12, torch.float16)):
        qkv = torch.randn(B, N, 3 * nh * 64, generator=g).cuda().to(dt)
        ms = ev(lambda: ops.seq_attention(qkv, nh, 0.125))
        flop = 4.0 * N * N * 64 * nh * B
        nbytes = B * N * nh * 64 * 4 * 2.0
        print(f"impl={os.environ.get('S6D_SEQ_ATTN_IMPL', '2')} {name}: B={B} N={N} heads={nh}: {ms * 1e3:.1f} us/launch  "
              f"{flop / ms / 1e9:.0f} TFLOP/s ({flop / ms / 1e9 / 2500:.3f} of the matrix peak)  {nbytes / ms / 1e6:.0f} GB/s "
              f"({nbytes / ms / 1e6 / 8000:.3f} of HBM)", flush=True)
