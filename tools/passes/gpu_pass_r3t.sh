# fused linear-attention tail: parity (PEM goldens), PEM stage, library-op profile
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plin.py tests/test_gpu_pem.py tests/test_gpu_zz_frame.py -x -q 2>&1 | tail -4 > $O/1_tests.txt
S6D_DISABLE_FUSED=linear_attention timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/2_bench_off.json 2> $O/2_bench_off.err
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_on.json 2> $O/3_bench_on.err
timeout 300 python tools/pem_ops_profile.py 32 2>/dev/null | grep -v "Warning\|warn" > $O/4_pem_ops.txt
cat $O/1_tests.txt
python - <<'PY'
import json
for f in ("2_bench_off", "3_bench_on"):
    try:
        d = json.loads(open(f"gpurun_out/r3t/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["stages_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
head -30 $O/4_pem_ops.txt
