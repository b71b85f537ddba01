cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sam_decoder.py tests/test_gpu_zz_pipeline.py -x -q 2>&1 | tail -2
bash tools/prof_samdec.sh > /dev/null 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof/samdec_v2_kernel_stats.csv")))
for r in rows[:8]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
timeout 600 python - <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import frame_demo
d = frame_demo.measure(torch.device("cuda", 0))
print({k: d[k] for k in ("frames_per_s", "ms_per_frame", "ms_per_frame_in_groups_of_8", "stages_ms")})
PY
