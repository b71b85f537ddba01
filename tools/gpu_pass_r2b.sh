# Round 2, pass b: state of the tree on the device after the re-entry -- whole GPU suite, the PEM pre-processing kernels that
# had no device run in round 1 (S6D_PEM_PRE=kernels), per-stage timing of that path, and the bench A/B with / without the
# hand-written GEMM.     gpurun --timeout 1500 -- 'bash tools/gpu_pass_r2b.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/1_gpu_suite.txt
S6D_PEM_PRE=kernels S6D_PEM_SEQ_CENTROID=1 S6D_PEM_SAMPLER=kernel timeout 400 python -m pytest tests/test_gpu_zz_host_glue.py tests/test_gpu_pem_pre.py -q -m gpu 2>&1 | tail -15 > $O/2_pem_pre_kernels.txt
timeout 200 python tools/pem_pre_time.py 64 > $O/2_pem_pre_stages.txt 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/3_bench_gemm.json 2> $O/3.err
S6D_DISABLE_FUSED=gemm_bf16 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/4_bench_library.json 2> $O/4.err
for f in $O/1_*.txt $O/2_*.txt; do echo "== $f"; tail -12 $f; done
for f in $O/3_bench_gemm.json $O/4_bench_library.json; do echo "== $f"; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('roofline'))"; done
tail -3 $O/3.err $O/4.err
