cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
rm -f gpurun_out/margins.jsonl
timeout 2400 python -m pytest tests/test_gpu_pose.py tests/test_gpu_pem.py tests/test_gpu_pem_pre.py tests/test_gpu_zz_frame.py tests/test_gpu_zz_pipeline.py tests/test_gpu_zz_sharded.py tests/test_gpu_zz_pipeline_e2e.py -q 2>&1 | tail -60 > gpurun_out/r5c/tests.txt
cp gpurun_out/margins.jsonl gpurun_out/r5c/margins.jsonl 2>/dev/null
tail -40 gpurun_out/r5c/tests.txt
