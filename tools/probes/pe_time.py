"""PositionalEncoding's fused kernel (s6d_pe_group_mlp_f32) timed on the bench's shapes: B clouds of 2048 points, the two
(radius, nsample) scales of fine_point_matching.py:90-99, HIP events on the launch stream.  Usage: python tools/probes/pe_time.py [B ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sam6d_amd import ops  # noqa: E402
from sam6d_amd.pem.pose_estimation_model import PositionalEncoding  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402


def event_ms(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    pe = seeded.load_seeded(PositionalEncoding(256).eval(), 6).cuda()
    for B in [int(a) for a in sys.argv[1:]] or [32, 10]:
        inp = synth.pem_inputs(B, seed=9, with_rgb=False)
        pts = (inp["dense_po"] / (inp["dense_po"].norm(dim=2).max(1)[0].reshape(-1, 1, 1) + 1e-6)).cuda().contiguous()
        for mlp, (r, ns) in zip((pe.mlp1, pe.mlp2), pe.scales):
            idx = ops.ball_query(pts, pts, r, ns)
            ws = [t for layer in mlp.layers() for t in layer.folded()]
            ms = event_ms(lambda: ops.pe_group_mlp(pts, idx, *ws))
            flop = 2.0 * B * 2048 * ns * (6 * 32 + 32 * 64 + 64 * 128)
            print(f"B={B} nsample={ns}: {ms:.4f} ms  ({flop / ms / 1e9:.1f} TFLOP/s as written, x3 executed on the bf16 cores for layers 1-2)", flush=True)
        with torch.no_grad():
            print(f"B={B} whole PositionalEncoding.forward: {event_ms(lambda: pe(pts)):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
