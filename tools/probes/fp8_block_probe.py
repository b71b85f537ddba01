"""What fp8 (e4m3) GEMM operands would cost in accuracy on ONE ViT-H block (BASELINE configs[4] decision aid; runs on the host):
oracle.sam.block in fp32 against the same block with the operands of its four Linear layers quantised -- bf16 (the path of
configs[1]), fp8 with one scale per tensor, fp8 with one scale per row (activations: per token, weights: per output channel).
Prints the rms error of the block's output relative to the output rms, and of the MLP branch alone.
usage: python tools/probes/fp8_block_probe.py [block index]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import sam as osam  # noqa: E402
from sam6d_amd.sam.image_encoder import build_vit_h  # noqa: E402
from sam6d_amd.utils import seeded  # noqa: E402

FP8_MAX = 448.0


def q_bf16(x, dim=None):
    return x.to(torch.bfloat16).float()


def q_fp8_tensor(x, dim=None):
    s = x.abs().max().clamp(min=1e-12) / FP8_MAX
    return (x / s).to(torch.float8_e4m3fn).float() * s


def q_fp8_row(x, dim=-1):
    s = x.abs().amax(dim=dim, keepdim=True).clamp(min=1e-12) / FP8_MAX
    return (x / s).to(torch.float8_e4m3fn).float() * s


def main():
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    m = seeded.load_seeded(build_vit_h().eval(), 3)
    blk = m.blocks[index]
    name = f"blocks.{index}"
    W = {f"{name}.{k}": v.float() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(40 + index)
    x = 0.5 * torch.randn(1, 64, 64, 1280, generator=g)
    real_linear = F.linear
    with torch.no_grad():
        ref = osam.block(W, name, x, 16, blk.window_size)
        for tag, q in (("bf16 operands", q_bf16), ("fp8 e4m3, per-tensor scale", q_fp8_tensor), ("fp8 e4m3, per-row scale", q_fp8_row)):
            def qlinear(inp, w, b=None):
                if w.shape[1] in (1280, 5120) and w.dim() == 2:          # the four Linear layers of the block
                    return real_linear(q(inp, -1), q(w, -1), b)
                return real_linear(inp, w, b)
            F.linear = qlinear
            try:
                out = osam.block(W, name, x, 16, blk.window_size)
            finally:
                F.linear = real_linear
            err = (out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()
            br = ((out - x) - (ref - x)).pow(2).mean().sqrt() / (ref - x).pow(2).mean().sqrt()
            print(f"block {index}: {tag:30s} rms error / output rms = {err:.2e}   (residual branches alone: {br:.2e})", flush=True)


if __name__ == "__main__":
    main()
