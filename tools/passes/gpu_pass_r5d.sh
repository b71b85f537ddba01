cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
rm -f gpurun_out/margins.jsonl
timeout 2400 python -m pytest tests/test_gpu_zz_pipeline.py tests/test_gpu_zz_sharded.py tests/test_gpu_zz_pipeline_e2e.py -q 2>&1 > gpurun_out/r5d/tests_full.txt
cp gpurun_out/margins.jsonl gpurun_out/r5d/margins.jsonl 2>/dev/null
grep -n "^E  \|Error\|passed\|failed" gpurun_out/r5d/tests_full.txt | cut -c1-400 | head -60
