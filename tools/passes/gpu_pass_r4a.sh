# round 4, first GPU call: the suite at HEAD (incl. the T-LESS / YCB-V goldens), the bench line, the whole-frame demo under rocprofv3 at HEAD
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r04
O=gpurun_out/prof; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/${R}_gpu_suite_first.txt
cat $O/${R}_gpu_suite_first.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/${R}_bench_first.json 2> $O/${R}_bench_first.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/prof/r04_bench_first.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "stages_ms", "roofline", "extras_error")})
print("pipeline", d.get("pipeline"))
for k in d.get("kernels", []):
    print(k["kernel"][:90], k["avg_ms"], k["frac"])
PY
cat > /tmp/fd.py <<'PY'
import json, sys, torch
sys.path.insert(0, "tools")
import frame_demo
d = frame_demo.measure(torch.device("cuda", 0))
print(json.dumps(d))
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fd -o fd -- python /tmp/fd.py > $O/${R}_frame_demo_first.txt 2>/dev/null
cp $(find /tmp/prof_fd -name "*kernel_stats.csv" | head -1) $O/${R}_frame_demo_kernel_stats_first.csv
head -40 $O/${R}_frame_demo_kernel_stats_first.csv | cut -c1-160
