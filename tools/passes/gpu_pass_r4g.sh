# round 4: decoder batch as a hipGraph (test + whole-frame stage times), the bench line with configs.fp8 and the range guard in the step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_sam_decoder.py tests/test_gpu_zz_pipeline.py tests/test_gpu_zz_frame.py "tests/test_gpu_pem.py::test_fp16_extractor_range_guard_reruns_overflowing_instances_in_fp32" -q 2>&1 | tail -15
cat > /tmp/fd.py <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import frame_demo
d = frame_demo.measure(torch.device("cuda", 0))
print(json.dumps({k: d[k] for k in ("frames_per_s", "ms_per_frame", "ms_per_frame_in_groups_of_8", "stages_ms")}))
PY
S6D_PEM_VIT_DTYPE=fp16 timeout 600 python /tmp/fd.py 2>&1 | tail -1
S6D_AMG_GRAPH=0 S6D_PEM_VIT_DTYPE=fp16 timeout 600 python /tmp/fd.py 2>&1 | tail -1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_g.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stages_ms", "extras_error")})
c = d.get("configs", {}).get("fp8", {})
print("fp8", {k: c.get(k) for k in ("value", "ms_per_step", "sam_encoder_ms", "roofline", "error")})
PY
tail -3 gpurun_out/bench_g.err
