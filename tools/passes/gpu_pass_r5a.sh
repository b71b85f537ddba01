# round 5, first GPU call: the FramePipeline pixels-to-pose tests against the reference-made golden, the group-vs-single probe,
# and a short bench for this box's baseline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_zz_pipeline_e2e.py -x -q 2>&1 | tail -40 > gpurun_out/r5a/e2e.txt
cp gpurun_out/margins.jsonl gpurun_out/r5a/margins_e2e.jsonl 2>/dev/null
timeout 600 python tools/probes/group_exact.py > gpurun_out/r5a/group_exact.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
tail -30 gpurun_out/r5a/e2e.txt; cat gpurun_out/r5a/group_exact.txt | tail -20; cut -c1-600 gpurun_out/r5a/bench.json
