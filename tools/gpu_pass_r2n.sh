# Round 2, pass n: batched ISM scoring (parity test + bench A/B with S6D_ISM_CHUNK=0)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ism.py -q -m gpu 2>&1 | tail -4 > $O/1_tests.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/2_bench.json 2> $O/2.err
S6D_ISM_CHUNK=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_perframe.json 2> $O/3.err
cat $O/1_tests.txt
for f in $O/2_bench.json $O/3_bench_perframe.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; done
tail -n 3 $O/2.err
