cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
rm -f gpurun_out/margins.jsonl
timeout 600 python tools/attn_time.py 16 > gpurun_out/r5g/attn_time.txt 2>&1
cp gpurun_out/attn_time.json gpurun_out/r5g/ 2>/dev/null
timeout 2400 python -m pytest tests/test_gpu_attn.py tests/test_gpu_sam.py tests/test_gpu_zz_pipeline.py tests/test_gpu_zz_sharded.py tests/test_gpu_zz_pipeline_e2e.py -q > gpurun_out/r5g/tests_full.txt 2>&1
cp gpurun_out/margins.jsonl gpurun_out/r5g/margins.jsonl 2>/dev/null
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err
grep -v seq257 gpurun_out/r5g/attn_time.txt | cut -c1-250
grep -n "^E  \|Error\|passed\|failed" gpurun_out/r5g/tests_full.txt | cut -c1-300 | head -40
cut -c1-400 gpurun_out/r5g/bench.json
