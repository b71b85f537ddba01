"""RPE attention core (s6d_rpe_attention_f32) and the plain multi-head rows kernel (s6d_mha_f32) per build of csrc/s6d_rpe.hip
under tools/rpe_variants/ (librpe_<name>.so), one process: time (HIP events) and bit-equality of the outputs with the first build.
Usage: python tools/probes/rpe_ab.py [B ...]"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vp = ctypes.c_void_p


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
    libs = sorted(glob.glob(os.path.join(ROOT, "tools", "rpe_variants", "librpe_*.so")))
    N, C = 197, 256
    for B in [int(a) for a in sys.argv[1:]] or [32, 10]:
        g = torch.Generator(device="cuda").manual_seed(B)
        q, k, v = (torch.randn(B, N, C, generator=g, device="cuda") for _ in range(3))
        qt = torch.randn(B, 4, N, C, generator=g, device="cuda") * 0.1
        qb = torch.randn(B, 4, N, generator=g, device="cuda")
        emb = torch.randn(B, N, N, C, generator=g, device="cuda")
        ref = {}
        for rep in range(2):
            for path in libs:
                name = os.path.basename(path)[7:-3]
                L = ctypes.CDLL(path)
                out = torch.empty(B, N, C, device="cuda")
                out2 = torch.empty(B, N, C, device="cuda")
                st = vp(torch.cuda.current_stream().cuda_stream)

                def rpe():
                    rc = L.s6d_rpe_attention_f32(vp(q.data_ptr()), vp(k.data_ptr()), vp(v.data_ptr()), vp(qt.data_ptr()), vp(qb.data_ptr()),
                                                 vp(emb.data_ptr()), B, N, C, 4, ctypes.c_float(0.125), vp(out.data_ptr()), st)
                    assert rc == 0, rc

                def mha():
                    rc = L.s6d_mha_f32(vp(q.data_ptr()), vp(k.data_ptr()), vp(v.data_ptr()), B, N, N, C, 4, ctypes.c_float(0.125),
                                       vp(out2.data_ptr()), st)
                    assert rc == 0, rc
                t1, t2 = event_ms(rpe), event_ms(mha)
                gb = emb.numel() * 4 / 1e9
                if "rpe" not in ref:
                    ref["rpe"], ref["mha"] = out.clone(), out2.clone()
                print(f"B={B} round {rep} {name:10s} rpe {t1:.4f} ms ({gb / t1:.2f} TB/s on the embedding)  mha {t2:.4f} ms  "
                      f"equal to the first build: {torch.equal(out, ref['rpe'])} / {torch.equal(out2, ref['mha'])}", flush=True)


if __name__ == "__main__":
    main()
