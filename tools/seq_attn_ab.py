"""A/B of s6d_seq_attention_bf16 between stand-alone builds of csrc/s6d_attn.hip (tools/attn_variants/libattn_<name>.so), ONE process:
DINOv2 ViT-L (crops x 16 heads x 257 tokens x 64) and the PEM ViT-B shape (32 x 12 x 197).  Outputs must be bit-identical.
usage: seq_attn_ab.py [crops]"""
import ctypes
import glob
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vp = ctypes.c_void_p


def ev(fn, n=30):
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
    libs = sorted(glob.glob(os.path.join(ROOT, "tools", "attn_variants", "libattn_*.so")), key=lambda p: "prev" not in p)
    g = torch.Generator(device="cuda").manual_seed(1)
    rows = {}
    for shape, (B, N, nh) in (("dinov2_%d" % crops, (crops, 257, 16)), ("pem_vitb_32", (32, 197, 12)), ("pem_vitb_10", (10, 197, 12))):
        qkv = torch.randn(B, N, 3 * nh * 64, generator=g, device="cuda").to(torch.bfloat16)
        ref = None
        for path in libs:
            name = os.path.basename(path)[8:-3]
            L = ctypes.CDLL(path)
            out = torch.empty(B, N, nh * 64, dtype=torch.bfloat16, device="cuda")

            def run():
                rc = L.s6d_seq_attention_bf16(vp(qkv.data_ptr()), B, N, nh, 64, ctypes.c_float(0.125), vp(out.data_ptr()),
                                              vp(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            ms = min(ev(run) for _ in range(3))
            row = {"us": round(ms * 1e3, 1), "tflops": round(4.0 * B * nh * 64 * N * N / ms / 1e9, 1),
                   "hbm_TBps": round((qkv.numel() + out.numel()) * 2 / ms / 1e9, 2)}
            if ref is None:
                ref = out.clone()
            else:
                row["equals_first"] = bool(torch.equal(out, ref))
            rows[f"{shape}.{name}"] = row
            print(f"{shape:14s} {name:10s} {json.dumps(row)}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "seq_attn_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
