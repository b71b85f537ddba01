# LN-fold pass: parity of the new GEMM forms, per-launch costs, bench A/B
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_sam.py -x -q 2>&1 | tail -5 > $O/1_tests.txt
timeout 300 python tools/probes/lnfold_micro.py > $O/2_micro.txt 2>&1
S6D_LNFOLD=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_off.json 2> $O/3_bench_off.err
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/4_bench_on.json 2> $O/4_bench_on.err
cp gpurun_out/margins.jsonl $O/margins.jsonl
cat $O/1_tests.txt; grep -v amdgpu.ids $O/2_micro.txt
python - <<'PY'
import json
for f in ("3_bench_off", "4_bench_on"):
    try:
        d = json.loads(open(f"gpurun_out/r3j/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["stages_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/4_bench_on.err
