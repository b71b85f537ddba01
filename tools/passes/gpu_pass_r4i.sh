# round 4: MX forms of the fp8 GEMM (lin1 -> MX e4m3 -> lin2): device tests, accuracy gate of both fp8 modes, SAM stage time per mode
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_fp8.py -q -k "mx or gate" 2>&1 | tail -12
grep "gate" gpurun_out/margins.jsonl
timeout 600 python - <<'PY' 2>&1 | grep -v Warn | tail -8
import os, sys, torch
sys.path.insert(0, ".")
import bench
hp = bench.HotPath(torch.device("cuda", 0), 32, 16)
from sam6d_amd import ops
for mode in ("bf16", "fp8", "fp8mx", "fp8", "fp8mx"):
    os.environ["S6D_SAM_GEMM"] = mode
    ms = bench.stage_ms(hp.sam_stage, 2)
    print(f"SAM stage, 32 frames, S6D_SAM_GEMM={mode}: {ms:.2f} ms", flush=True)
os.environ["S6D_SAM_GEMM"] = "fp8mx"
M = 65536
from sam6d_amd.utils import fp8
g = torch.Generator().manual_seed(0)
qa, sa = fp8.quantize_rows(torch.randn(M, 1280, generator=g).cuda())
qw, sw = fp8.quantize_rows((torch.randn(5120, 1280, generator=g) / 36).cuda())
b = torch.randn(5120, generator=g).cuda()
print("lin1 fp8 + GELU -> bf16 :", round(bench._event_ms(lambda: ops.gemm_fp8(qa, sa, qw, sw, b, gelu=True), 10), 4), "ms")
print("lin1 fp8 + GELU -> MX   :", round(bench._event_ms(lambda: ops.gemm_fp8_gelu_mx(qa, sa, qw, sw, b), 10), 4), "ms")
q8, qs = ops.gemm_fp8_gelu_mx(qa, sa, qw, sw, b)
qw2, sw2 = fp8.quantize_rows((torch.randn(1280, 5120, generator=g) / 72).cuda())
b2 = torch.randn(1280, generator=g).cuda()
print("lin2 fp8 MX -> bf16     :", round(bench._event_ms(lambda: ops.gemm_fp8_mxa(q8, qs, qw2, sw2, b2), 10), 4), "ms")
h = ops.gemm_fp8(qa, sa, qw, sw, b, gelu=True)
w2 = (torch.randn(1280, 5120, generator=g) / 72).cuda().to(torch.bfloat16)
xr = torch.randn(M, 1280, generator=g).cuda().to(torch.bfloat16)
print("lin2 bf16 + residual    :", round(bench._event_ms(lambda: ops.gemm_bf16(h, w2, b2, residual=xr, out=xr), 10), 4), "ms")
PY
