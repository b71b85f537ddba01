# descriptor ViT batched across the frames of a group; ln_stats_finalize after the latency fix
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zz_pipeline.py tests/test_gpu_dinov2.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -4 > $O/1_tests.txt
timeout 300 python tools/probes/lnfold_micro.py > $O/2_micro.txt 2>&1
timeout 600 python - > $O/3_pipeline.json 2> $O/3_pipeline.err <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import frame_demo
print(json.dumps(frame_demo.measure(torch.device("cuda", 0))))
PY
cat $O/1_tests.txt; grep -v amdgpu.ids $O/2_micro.txt | tail -4; cat $O/3_pipeline.json; tail -3 $O/3_pipeline.err
