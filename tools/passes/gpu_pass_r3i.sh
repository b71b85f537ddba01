# Round 3, pass i: IEEE-half ViT-B pipeline of the PEM (first device run): kernel tests, pose parity, bench
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_attn.py tests/test_gpu_pem.py tests/test_gpu_dinov2.py tests/test_gpu_zz_pipeline.py -q -m gpu 2>&1 | tail -25 > $O/1_tests.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl 2>/dev/null
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/2_bench.json 2> $O/2.err
cat $O/1_tests.txt; cat $O/margins.jsonl
python -c "
import json; d=json.loads(open('$O/2_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('pem_vit_dtype'), d.get('extras_error'))"
