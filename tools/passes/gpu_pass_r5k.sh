cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_zz_frame_e2e_nms.py -q > gpurun_out/r5k/nms.txt 2>&1
grep -n "^E  \|Error\|passed\|failed" gpurun_out/r5k/nms.txt | cut -c1-400 | head; cat gpurun_out/margins.jsonl 2>/dev/null | cut -c1-300
timeout 900 python - <<'PY' > gpurun_out/r5k/sharded_world1.json 2> gpurun_out/r5k/sharded.err
import json, os, sys, torch
os.environ.setdefault("S6D_PEM_VIT_DTYPE", "fp16")
sys.path.insert(0, ".")
from tools import run_sharded
print(json.dumps(run_sharded.measure_world1(torch.device("cuda", 0))))
PY
cat gpurun_out/r5k/sharded_world1.json | cut -c1-600; tail -3 gpurun_out/r5k/sharded.err | cut -c1-300
