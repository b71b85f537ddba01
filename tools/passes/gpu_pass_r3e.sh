# Round 3, pass e: fused fp32 Linear of the PEM (first device run), PEM stage A/B, fp32 ViT-B cost
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_plin.py tests/test_gpu_pose.py tests/test_gpu_pem.py tests/test_gpu_fp8.py tests/test_gpu_sam.py -q -m gpu 2>&1 | tail -25 > $O/1_tests.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl 2>/dev/null
timeout 300 python - > $O/2_plin_micro.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
from sam6d_amd import ops
def ms(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
for M in (65536, 6304):
    for K, N in ((256, 256), (256, 512), (512, 256), (256, 768)):
        x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** .5).cuda(); b = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).cuda(); gm, bt = torch.ones(N).cuda(), torch.zeros(N).cuda()
        hi, lo = ops.split_weight(w)
        t_k = ms(lambda: ops.linear_f32(x, hi, lo, b))
        t_l = ms(lambda: torch.nn.functional.linear(x, w, b))
        line = f"M={M} K={K} N={N}: kernel {t_k*1e3:.1f} us ({2*M*N*K*3/t_k/1e9:.0f} TFLOP/s bf16 executed), library fp32 {t_l*1e3:.1f} us"
        if N == 256:
            t_kf = ms(lambda: ops.linear_f32(x, hi, lo, b, residual=r, ln=(gm, bt, 1e-5)))
            t_lf = ms(lambda: torch.nn.functional.layer_norm(torch.nn.functional.linear(x, w, b) + r, (N,), gm, bt, 1e-5))
            line += f"; + residual + LayerNorm: kernel {t_kf*1e3:.1f} us, library {t_lf*1e3:.1f} us"
        print(line, flush=True)
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench.json 2> $O/3.err
S6D_DISABLE_FUSED=linear_f32 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/4_bench_noplin.json 2> $O/4.err
S6D_PEM_VIT_DTYPE=fp32 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/5_bench_fp32vit.json 2> $O/5.err
cat $O/1_tests.txt; grep -v amdgpu.ids $O/2_plin_micro.txt
for f in $O/3_bench.json $O/4_bench_noplin.json $O/5_bench_fp32vit.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('extras_error'))"; done
