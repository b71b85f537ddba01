"""Feasibility probe for BASELINE configs[4] (fp8 ViT-H path): does torch._scaled_mm run on this stack, and how fast is
the ViT-H MLP GEMM (M = 65536, K = 1280, N = 5120) in OCP e4m3 against bf16?  (run on the GPU box)"""
import torch


def ev(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


M, K, N = 65536, 1280, 5120
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda()
w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
xb, wb = x.bfloat16(), w.bfloat16()
ref = (xb[:256].float() @ wb.float().t())
ms = ev(lambda: torch.nn.functional.linear(xb, wb))
print(f"bf16 linear: {ms:.3f} ms  {2 * M * K * N / ms / 1e9:.0f} TFLOP/s", flush=True)
for dt in ("float8_e4m3fn", "float8_e4m3fnuz"):
    try:
        f8 = getattr(torch, dt)
        sx, sw = x.abs().max() / 448.0, w.abs().max() / 448.0
        x8, w8 = (x / sx).to(f8), (w / sw).to(f8)
        one = torch.ones((), device="cuda")
        fn = lambda: torch._scaled_mm(x8, w8.t(), scale_a=sx.reshape(()).float(), scale_b=sw.reshape(()).float(),  # noqa: E731
                                      out_dtype=torch.bfloat16)
        y = fn()
        err = (y[:256].float() - ref).abs().mean() / ref.abs().mean()
        ms = ev(fn)
        print(f"{dt} _scaled_mm: {ms:.3f} ms  {2 * M * K * N / ms / 1e9:.0f} TFLOP/s  rel.err vs bf16 {err:.3e}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{dt}: {type(e).__name__}: {str(e)[:200]}", flush=True)
