# scalar-bias A/B, LN-fold pass 3 (async row statistics, scalar column constants)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -3 > $O/1_tests.txt
timeout 300 python tools/probes/lnfold_micro.py > $O/2_micro.txt 2>&1
S6D_LNFOLD=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_off.json 2> $O/3_bench_off.err
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/4_bench_on.json 2> $O/4_bench_on.err
python - <<'PY'
import json
for f in ("3_bench_off", "4_bench_on"):
    try:
        d = json.loads(open(f"gpurun_out/r3s/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["stages_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
