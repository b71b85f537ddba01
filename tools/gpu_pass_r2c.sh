# Round 2, pass c: GEMM version 2 (two independent 256 x 128 workgroups per CU) against version 1 and hipBLASLt, the fused
# fine-matching kernels (parity + timing), bench.     gpurun --timeout 1200 -- 'bash tools/gpu_pass_r2c.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_pose.py tests/test_gpu_pem.py -x -q -m gpu 2>&1 | tail -12 > $O/1_tests.txt
S6D_GEMM_IMPL=2 timeout 300 python tools/gemm_time.py shapes > $O/2_gemm_impl2.txt 2>&1
S6D_GEMM_IMPL=1 timeout 300 python tools/gemm_time.py shapes > $O/2_gemm_impl1.txt 2>&1
timeout 200 python tools/fine_time.py 32 > $O/3_fine_time.txt 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/4_bench.json 2> $O/4.err
cp gpurun_out/gemm_time_impl*.json gpurun_out/fine_time.json $O/ 2>/dev/null
echo "== tests"; cat $O/1_tests.txt
echo "== impl2"; grep -v amdgpu.ids $O/2_gemm_impl2.txt | cut -c1-230
echo "== impl1"; grep -v amdgpu.ids $O/2_gemm_impl1.txt | cut -c1-230
echo "== fine"; tail -3 $O/3_fine_time.txt
echo "== bench"; python -c "import json,sys; d=json.loads(open('$O/4_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; tail -n 3 $O/4.err
