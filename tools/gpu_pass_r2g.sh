# Round 2, pass g: GEMM (quad-transposed epilogue) against hipBLASLt in one process, bench A/B incl. the whole-frame pipeline block
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 300 python tools/gemm_time.py shapes > $O/1_gemm_shapes.txt 2>&1; cp gpurun_out/gemm_time_impl*.json $O/
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/2_bench.json 2> $O/2.err
S6D_DISABLE_FUSED=gemm_bf16 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > $O/3_bench_library.json 2> $O/3.err
echo "== gemm"; grep -v amdgpu.ids $O/1_gemm_shapes.txt | cut -c1-250
for f in $O/2_bench.json $O/3_bench_library.json; do echo "== $f"; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('roofline'), d.get('pipeline'))"; done
tail -n 5 $O/2.err
