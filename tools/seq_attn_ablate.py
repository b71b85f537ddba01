"""Times s6d_seq_attention_bf16 of every tools/attn_variants/libattn_seq_*.so at the DINOv2 shape (150 crops x 16 heads x 257 tokens)."""
import ctypes
import glob
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, N, nh = 150, 257, 16
qkv = torch.randn(B, N, 3 * nh * 64, generator=torch.Generator().manual_seed(0)).cuda().to(torch.bfloat16)
out = torch.empty(B, N, nh * 64, dtype=torch.bfloat16, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for impl in (sys.argv[1] if len(sys.argv) > 1 else "2",):       # the choice is a per-library static read at the first call: one process per impl
    os.environ["S6D_SEQ_ATTN_IMPL"] = impl
    for so in sorted(glob.glob(os.path.join(root, "tools", "attn_variants", "libattn_seq_*.so"))):
        L = ctypes.CDLL(so)
        f = L.s6d_seq_attention_strided_bf16
        f.restype = ctypes.c_int
        hm = len(sys.argv) > 2 and sys.argv[2] == "head"        # head-major (3 nh, B N, 64) instead of token-major (B, N, 3 nh 64)
        ts, ws, hs = (64, nh * B * N * 64, B * N * 64) if hm else (3 * nh * 64, nh * 64, 64)

        def run():
            rc = f(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_long(ts), ctypes.c_long(ws), ctypes.c_long(hs), B, N, nh, 64,
                   ctypes.c_float(0.125), ctypes.c_void_p(out.data_ptr()), st)
            assert rc == 0, rc
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"impl={impl} {'head ' if hm else 'token'} {os.path.basename(so)[8:-3]:>16}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us/launch", flush=True)
        # a fresh library per variant: the impl choice is a function-local static of each library, read at its first call
