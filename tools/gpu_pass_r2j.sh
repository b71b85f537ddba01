# Round 2, pass j: window attention, two workgroups per item (S6D_WIN16_IMPL=2, default) against one (=1), parity tests, bench
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_attn.py tests/test_gpu_sam.py -q -m gpu 2>&1 | tail -5 > $O/1_tests.txt
for impl in 2 1 2 1; do S6D_WIN16_IMPL=$impl timeout 120 python -c "
import sys; sys.path.insert(0, '.')
from tools.attn_ablate import run
run(16, 64, 16, 80, 14, $impl, n=30)" 2>&1 | grep -v amdgpu.ids >> $O/2_win16_time.txt; done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench.json 2> $O/3.err
cat $O/1_tests.txt $O/2_win16_time.txt
python -c "import json,sys; d=json.loads(open('$O/3_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms')); print([ (k['kernel'][:40], k['avg_ms']) for k in d['kernels']])"
