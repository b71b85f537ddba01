# Round 3, pass c: fp8 kernels (first device run), residual epilogue after the asm-DMA change, bench lmo / fp8
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_gemm.py tests/test_gpu_sam.py -q -m gpu -x 2>&1 | tail -25 > $O/1_tests.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl 2>/dev/null
timeout 300 python - > $O/2_gemm_micro.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
from sam6d_amd import ops
from sam6d_amd.utils import fp8
def ms(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
M = 65536
for nm, K, N in (("proj", 1280, 1280), ("lin2", 5120, 1280)):
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16); w = (torch.randn(N, K, generator=g) / K ** .5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda(); x = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16)
    for rep in range(2):
        print(f"{nm}: plain {ms(lambda: ops.gemm_bf16(a, w, b)):.4f} ms, residual epilogue {ms(lambda: ops.gemm_bf16(a, w, b, residual=x, out=x)):.4f} ms", flush=True)
gm, bt = torch.ones(1280).cuda(), torch.zeros(1280).cuda()
xb = torch.randn(M, 1280, generator=g).cuda().to(torch.bfloat16)
print(f"add_layernorm 2-read {ms(lambda: ops.add_layernorm(xb, xb, gm, bt, 1e-6)):.4f} ms, 1-read {ms(lambda: ops.add_layernorm(xb, None, gm, bt, 1e-6)):.4f} ms, layernorm_fp8 {ms(lambda: ops.layernorm_fp8(xb, gm, bt, 1e-6)):.4f} ms")
for nm, K, N, gelu in (("qkv", 1280, 3840, False), ("lin1+gelu", 1280, 5120, True)):
    qa, sa = fp8.quantize_rows(torch.randn(M, K, generator=g).cuda()); qw, sw = fp8.quantize_rows((torch.randn(N, K, generator=g) / K ** .5).cuda())
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16); w = (torch.randn(N, K, generator=g) / K ** .5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    for rep in range(2):
        t8, t16 = ms(lambda: ops.gemm_fp8(qa, sa, qw, sw, b, gelu=gelu)), ms(lambda: ops.gemm_bf16(a, w, b, gelu=gelu))
        print(f"{nm}: fp8 {t8:.4f} ms = {2*M*N*K/t8/1e9:.0f} TFLOP/s, bf16 {t16:.4f} ms = {2*M*N*K/t16/1e9:.0f} TFLOP/s", flush=True)
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_lmo.json 2> $O/3.err
timeout 300 python bench.py --config fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/4_bench_fp8.json 2> $O/4.err
cat $O/1_tests.txt; cat $O/margins.jsonl; grep -v amdgpu.ids $O/2_gemm_micro.txt
for f in $O/3_bench_lmo.json $O/4_bench_fp8.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('roofline'), d.get('extras_error'))
for k in d.get('kernels', []): print('   ', k['kernel'], k.get('avg_ms'), k.get('achieved'), k.get('frac'))
"; done; tail -3 $O/4.err
