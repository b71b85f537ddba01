# Round 3, pass b: full GPU suite; bench A/B of the residual-add GEMM epilogue; run_sharded at world 1
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/1_tests.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/2_bench_res.json 2> $O/2.err
S6D_DISABLE_FUSED=gemm_bf16_res timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_nores.json 2> $O/3.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/4_bench_res.json 2> $O/4.err
timeout 600 python tools/run_sharded.py --frames 16 --group 8 --out $O/sharded.csv --fixed-time 0 > $O/5_sharded.json 2> $O/5.err
cat $O/1_tests.txt
for f in $O/2_bench_res.json $O/3_bench_nores.json $O/4_bench_res.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'), [(k['name'], k.get('avg_ms')) for k in d.get('kernels', [])][:12])"; done
cat $O/5_sharded.json; tail -3 $O/5.err; head -3 $O/sharded.csv
