cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
rm -f gpurun_out/margins.jsonl
timeout 600 python tools/probes/geo_half_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5i/geo_half_ab.txt
cat gpurun_out/r5i/geo_half_ab.txt
S6D_PEM_GEO_DTYPE=fp16 timeout 1500 python -m pytest tests/test_gpu_pem.py tests/test_gpu_zz_frame.py tests/test_gpu_zz_pipeline_e2e.py tests/test_host_example_frame.py -q -m gpu > gpurun_out/r5i/tests_geo_fp16.txt 2>&1
cp gpurun_out/margins.jsonl gpurun_out/r5i/margins_geo_fp16.jsonl; rm -f gpurun_out/margins.jsonl
grep -n "^E  \|Error\|passed\|failed" gpurun_out/r5i/tests_geo_fp16.txt | cut -c1-300 | head -20
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r5i/tests_all.txt 2>&1
cp gpurun_out/margins.jsonl gpurun_out/r5i/margins_all.jsonl
grep -n "^E  \|Error\|passed\|failed" gpurun_out/r5i/tests_all.txt | cut -c1-300 | head -30
