cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_pose.py tests/test_gpu_gemm.py tests/test_gpu_pem.py -q -x 2>&1 | tail -3
for ps in 1 0; do S6D_GEO_PRESPLIT=$ps timeout 300 python tools/probes/geo_half_ab.py 2>&1 | grep "geo fp32" | sed "s/^/presplit=$ps /"; done | tee gpurun_out/r5l/geo_presplit_ab.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-pipeline --no-fp8 --no-cpu-baseline > gpurun_out/r5l/bench.json 2> gpurun_out/r5l/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5l/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages_ms"])
for k in d.get("kernels", []):
    print(k["kernel"][:90], k["avg_ms"], k["frac"])
PY
