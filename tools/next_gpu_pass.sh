# First GPU call of the next round: device checks of everything written after the round-1 GPU budget ran out, then the two
# opt-in experiments, A/B against the default bench.  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/next_gpu_pass.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 1. new host-glue device tests first (short), then the whole suite
timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py tests/test_gpu_zz_pipeline.py -q 2>&1 | tail -5 > gpurun_out/n1_glue.txt
S6D_PEM_SEQ_CENTROID=1 timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py -q -k sequential 2>&1 | tail -5 > gpurun_out/n2_seq_centroid.txt
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/n3_gpu_suite.txt
# 2. the segmentor plugin end to end (ViT-H, seeded weights) and the five-model frame chain
timeout 600 python tools/segmentor_demo.py > gpurun_out/n4_segmentor_demo.txt 2>&1
timeout 600 python tools/frame_demo.py > gpurun_out/n4_frame_demo.txt 2>&1
# 3. bench A/B: default vs Infinity-Cache-sized MLP row chunks
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/n5_bench_default.json 2> gpurun_out/n5.err
S6D_SAM_MLP_ROWS=16384 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/n6_bench_mlp16384.json 2> gpurun_out/n6.err
S6D_SAM_MLP_ROWS=32768 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/n7_bench_mlp32768.json 2> gpurun_out/n7.err
for f in gpurun_out/n[1-4]*.txt; do echo "== $f"; cat $f | tail -6; done
for f in gpurun_out/n[5-7]*.json; do echo "== $f"; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; done
