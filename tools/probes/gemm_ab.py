"""A/B of stand-alone builds of csrc/s6d_gemm.hip (tools/probes/gemm_ab_build.sh: one library per -D setting) on the four ViT-H
Linear shapes at 16 frames, HIP events, interleaved rounds.  Usage: python tools/probes/gemm_ab.py name=path.so ..."""
import ctypes
import sys

import torch

vp = ctypes.c_void_p
MODES = (True,)
SHAPES = [("qkv", 1280, 3840, 0), ("proj", 1280, 1280, 0), ("lin1+gelu", 1280, 5120, 1), ("lin2", 5120, 1280, 0)]
M = 65536


def main():
    libs = {}
    for a in sys.argv[1:]:
        nm, path = a.split("=")
        libs[nm] = ctypes.CDLL(path)
    g = torch.Generator(device="cuda").manual_seed(0)
    st = vp(torch.cuda.current_stream().cuda_stream)
    res = {}
    xr = torch.randn(M, 1280, generator=g, device="cuda").to(torch.bfloat16)
    for rnd in range(3):
        for sn, K, N, gelu in SHAPES:
            a = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, generator=g, device="cuda")
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            for nm, L in libs.items():
                for bias in MODES:
                    if bias == "res" and (gelu or N != 1280):
                        continue

                    def fn():
                        if bias == "res":
                            rc = L.s6d_gemm_bf16_res(vp(a.data_ptr()), ctypes.c_long(K), vp(w.data_ptr()), ctypes.c_long(K), vp(b.data_ptr()),
                                                     vp(xr.data_ptr()), ctypes.c_long(N), vp(0), vp(xr.data_ptr()),
                                                     ctypes.c_long(N), M, N, K, 0, st)
                        else:
                            rc = L.s6d_gemm_bf16(vp(a.data_ptr()), ctypes.c_long(K), vp(w.data_ptr()), ctypes.c_long(K),
                                                 vp(b.data_ptr()) if bias else vp(0), vp(out.data_ptr()), ctypes.c_long(N), M, N, K, gelu, 0, st)
                        assert rc == 0, rc
                    fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault((sn, nm, bias), []).append(e0.elapsed_time(e1) / 20)
    print(f"# M = {M}, ms per launch (3 interleaved rounds of 20)")
    for (sn, nm, bias), v in res.items():
        print(f"{sn:10s} {nm:10s} {'residual' if bias == 'res' else ('bias' if bias else 'no bias'):8s} " + " ".join(f"{x:.4f}" for x in v))


main()
