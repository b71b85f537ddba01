"""Time s6d_gemm_bf16 (csrc/s6d_gemm.hip) against the library GEMM (hipBLASLt through torch) at the Linear shapes of the
three ViTs on the path, check it against an fp32 product of the same bf16 operands, and run the profiling variants built by
tools/gemm_variants.sh.  Usage: python tools/gemm_time.py [quick|full|shapes|pmc]   (writes gpurun_out/gemm_time.json)"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sam6d_amd import _lib, ops  # noqa: E402

vp = ctypes.c_void_p


def event_ms(fn, n=10, warm=2):
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


def call(L, a, w, b, out, gelu, max_blocks=0):
    M, K = a.shape
    N = w.shape[0]
    rc = L.s6d_gemm_bf16(vp(a.data_ptr()), ctypes.c_long(a.stride(0)), vp(w.data_ptr()), ctypes.c_long(w.stride(0)),
                         vp(b.data_ptr()) if b is not None else vp(0), vp(out.data_ptr()), ctypes.c_long(out.stride(0)),
                         M, N, K, 1 if gelu else 0, int(max_blocks), vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


def make(M, N, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda")
    return a, w, b


def check(a, w, b, out, gelu, rows=1024):
    idx = torch.randperm(a.shape[0], device="cuda")[:rows]
    ref = a[idx].float() @ w.float().t() + b
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = (out[idx].float() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-5
    return int((err > 1.01 * tol).sum().item()), float(err.max().item())


SHAPES = [  # name, M, K, N, gelu
    ("sam.qkv", 65536, 1280, 3840, False), ("sam.proj", 65536, 1280, 1280, False),
    ("sam.lin1+gelu", 65536, 1280, 5120, True), ("sam.lin2", 65536, 5120, 1280, False),
    ("sam.neck1x1", 65536, 1280, 256, False), ("sam.lin1+gelu M=32768", 32768, 1280, 5120, True),
    ("pemvit.qkv", 6304, 768, 2304, False), ("pemvit.fc1+gelu", 6304, 768, 3072, True), ("pemvit.fc2", 6304, 3072, 768, False),
    ("dino.qkv", 32896, 1024, 3072, False), ("dino.fc1+gelu", 32896, 1024, 4096, True), ("dino.fc2", 32896, 4096, 1024, False),
]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "full"
    L = _lib.lib()
    res = {"shapes": [], "variants": [], "grid": []}
    if mode == "pmc":                       # a few launches of the dominant shape only (counter passes)
        a, w, b = make(65536, 5120, 1280)
        out = torch.empty(65536, 5120, dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            call(L, a, w, b, out, True)
        torch.cuda.synchronize()
        return
    shapes = [] if mode == "variants" else SHAPES[:4] if mode == "quick" else SHAPES
    for name, M, K, N, gelu in shapes:
        a, w, b = make(M, N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        bb = b.to(torch.bfloat16)
        call(L, a, w, b, out, gelu)
        bad, emax = check(a, w, b, out, gelu)
        ms = event_ms(lambda: call(L, a, w, b, out, gelu))
        ms_lib = event_ms(lambda: torch.nn.functional.linear(a, w, bb))
        ms_gelu = 0.0
        if gelu:
            h = torch.nn.functional.linear(a, w, bb)
            ms_gelu = event_ms(lambda: torch.nn.functional.gelu(h))
        fl = 2.0 * M * N * K
        row = {"name": name, "M": M, "K": K, "N": N, "gelu": gelu, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1),
               "lib_ms": round(ms_lib, 4), "lib_tflops": round(fl / ms_lib / 1e9, 1), "lib_gelu_ms": round(ms_gelu, 4),
               "speedup_vs_lib_total": round((ms_lib + ms_gelu) / ms, 3), "mismatches": bad, "max_err": emax}
        res["shapes"].append(row)
        print(row, flush=True)
    if mode == "variants":                  # every library of tools/gemm_variants on the four ViT-H shapes
        import glob
        out = {}
        data = {name: (make(M, N, K), torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), gelu, 2.0 * M * N * K)
                for name, M, K, N, gelu in SHAPES[:4]}
        libs = sorted(glob.glob(os.path.join(ROOT, "tools", "gemm_variants", "libgemm_*.so")))
        for rep in range(2):                # two interleaved rounds: run-to-run spread next to the differences
            for path in libs:
                v = os.path.basename(path)[8:-3]
                Lv = ctypes.CDLL(path)
                for name, ((a, w, b), o, gelu, fl) in data.items():
                    ms = event_ms(lambda: call(Lv, a, w, b, o, gelu), n=20, warm=3)
                    out.setdefault(v, {}).setdefault(name, []).append(round(fl / ms / 1e9, 1))
        for v, d in out.items():
            print(v.ljust(12), {k: x for k, x in d.items()}, flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_variants.json"), "w"), indent=1)
        return
    if mode == "shapes":                    # the shape table only
        tag = "_shapes"
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"gemm_time{tag}.json"), "w"), indent=1)
        return
    # profiling variants, grid sizes and tile orders on the dominant shape
    a, w, b = make(65536, 5120, 1280)
    out = torch.empty(65536, 5120, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * 65536 * 5120 * 1280
    vdir = os.path.join(ROOT, "tools", "gemm_variants")
    for v in ("base", "noprio", "nodma", "nomfma", "nostore", "mfmaonly"):
        path = os.path.join(vdir, f"libgemm_{v}.so")
        if not os.path.exists(path):
            continue
        Lv = ctypes.CDLL(path)
        for gelu in (False, True):
            ms = event_ms(lambda: call(Lv, a, w, b, out, gelu))
            row = {"variant": v, "gelu": gelu, "ms": round(ms, 4), "tflops_equiv": round(fl / ms / 1e9, 1)}
            res["variants"].append(row)
            print(row, flush=True)
    for mb in (256, 512, 1024, 5120):
        ms = event_ms(lambda: call(L, a, w, b, out, True, mb))
        row = {"max_blocks": mb, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
        res["grid"].append(row)
        print(row, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_time.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
