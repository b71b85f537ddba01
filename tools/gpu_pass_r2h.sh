# Round 2, pass h: whole GPU suite with the new default paths (fused fine matching, coarse kernels, PEM pre-processing kernels),
# per-path timing of the pre-processing, bench
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/1_gpu_suite.txt
timeout 200 python tools/pem_pre_time.py 64 > $O/2_pem_pre.txt 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench.json 2> $O/3.err
echo "== suite"; cat $O/1_gpu_suite.txt
echo "== pempre"; tail -14 $O/2_pem_pre.txt
echo "== bench"; python -c "import json,sys; d=json.loads(open('$O/3_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; tail -n 3 $O/3.err
