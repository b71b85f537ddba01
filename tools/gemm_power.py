"""Board power and shader clock while the bf16 GEMM runs back to back (ours against hipBLASLt, lin1 shape 65536 x 1280 -> 5120 without
GELU): rocm-smi is polled from a thread during ~4 s of launches per arm.  Explains why 71 % matrix-pipe utilisation is ~1.0 PFLOP/s:
the chip runs these kernels at its power limit and the clock follows.   python tools/gemm_power.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sam6d_amd import ops  # noqa: E402


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            out.append({k: v for k, v in card.items() if "ower" in k or "sclk" in k.lower()})
        except Exception as e:  # noqa: BLE001
            out.append({"error": str(e)[:80]})
        time.sleep(0.25)


def arm(name, fn, flop, seconds=4.0):
    fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    print(json.dumps({"arm": name, "tflops": round(flop * n / dt / 1e12, 1), "samples": samples[1:-1][:12]}), flush=True)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    M, K, N = 65536, 1280, 5120
    a = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda")
    bb = b.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    flop = 2.0 * M * N * K
    arm("s6d_gemm_bf16", lambda: ops.gemm_bf16(a, w, b, out=out), flop)
    arm("hipBLASLt", lambda: torch.nn.functional.linear(a, w, bb), flop)
    # profiling variants (tools/gemm_variants.sh): which part of the kernel the power goes to
    import ctypes
    import glob
    vp = ctypes.c_void_p
    for path in sorted(glob.glob(os.path.join(ROOT, "tools", "gemm_variants", "libgemm_*.so"))):
        Lv = ctypes.CDLL(path)

        def call(Lv=Lv):
            rc = Lv.s6d_gemm_bf16(vp(a.data_ptr()), ctypes.c_long(a.stride(0)), vp(w.data_ptr()), ctypes.c_long(w.stride(0)), vp(b.data_ptr()),
                                  vp(out.data_ptr()), ctypes.c_long(out.stride(0)), M, N, K, 0, 0, vp(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        arm("variant " + os.path.basename(path)[8:-3], call, flop, 2.5)
    time.sleep(2)
    arm("idle-ish (1 launch / 10 ms)", lambda: (ops.gemm_bf16(a[:256], w, b), time.sleep(0.01)), 2.0 * 256 * N * K, 2.0)


if __name__ == "__main__":
    main()
