# Round 3, pass f: PEM stage after the no-concatenation dense loop: tests, stage time, rocprofv3 kernel stats of the stage (bf16 ViT-B)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pose.py tests/test_gpu_pem.py tests/test_gpu_zz_pipeline.py -q -m gpu 2>&1 | tail -8 > $O/1_tests.txt
timeout 300 python - > $O/2_pem_stage.txt 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, ".")
import bench
for dt in ("bf16", "fp32"):
    os.environ["S6D_PEM_VIT_DTYPE"] = dt
    for B in (32, 10):
        hp = bench.HotPath(torch.device("cuda", 0), B, min(B, 16))
        print(f"PEM stage, ViT-B {dt}, {B} instances: {bench.stage_ms(hp.pem_stage, 3):.2f} ms", flush=True)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pem -o pem -- python tools/run_stage.py pem 32 4 > /dev/null 2>&1
cp $(find /tmp/prof_pem -name "*kernel_stats.csv" | head -1) $O/r03_pem32_kernel_stats.csv
cat $O/1_tests.txt; grep -v amdgpu.ids $O/2_pem_stage.txt; head -45 $O/r03_pem32_kernel_stats.csv | cut -c1-200
