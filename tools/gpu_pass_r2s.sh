# Round 2, pass s: head-major q/k/v layout (GEMM column-block epilogue + attention strides): parity, kernel timings, bench A/B
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_attn.py tests/test_gpu_gemm.py tests/test_gpu_sam.py -q -m gpu 2>&1 | tail -5 > $O/1_tests.txt
timeout 200 python - > $O/2_attn_layout.txt 2>&1 <<'PY'
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
B, H, nh, hd = 16, 64, 16, 80
qkv = torch.randn(B, H, H, 3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
hm = qkv.view(B * H * H, 3 * nh, hd).permute(1, 0, 2).contiguous()
bias = torch.randn(3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
for ws, S in ((14, 14), (0, 64)):
    rh = torch.randn(2 * S - 1, hd, generator=g).cuda().to(torch.bfloat16); rw = torch.randn(2 * S - 1, hd, generator=g).cuda().to(torch.bfloat16)
    for rep in range(2):
        t = ms(lambda: ops.window_attention(qkv, bias, rh, rw, nh, ws, hd ** -0.5))
        h = ms(lambda: ops.window_attention(hm, bias, rh, rw, nh, ws, hd ** -0.5, head_major_shape=(B, H, H)))
        print(f"ws={ws}: token-major {t:.4f} ms, head-major {h:.4f} ms", flush=True)
a = torch.randn(B * H * H, 1280, generator=g).cuda().to(torch.bfloat16); w = (torch.randn(3840, 1280, generator=g) / 36).cuda().to(torch.bfloat16); b = torch.randn(3840, generator=g).cuda()
for rep in range(2):
    print(f"qkv GEMM: plain {ms(lambda: ops.gemm_bf16(a, w, b)):.4f} ms, column blocks {ms(lambda: ops.gemm_bf16(a, w, b, col_block=80)):.4f} ms", flush=True)
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_head.json 2> $O/3.err
S6D_QKV_LAYOUT=token timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/4_bench_token.json 2> $O/4.err
cat $O/1_tests.txt; grep -v amdgpu.ids $O/2_attn_layout.txt
for f in $O/3_bench_head.json $O/4_bench_token.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; done
