# round 4, last call: suite + smoke + the timed default bench line at HEAD (kernel traces and PMC summaries stay those of tools/final_pass.sh)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r04
mkdir -p gpurun_out/prof; rm -f gpurun_out/margins.jsonl
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/prof/${R}_gpu_suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/prof/${R}_gpu_suite.txt
cp gpurun_out/margins.jsonl gpurun_out/prof/${R}_parity_margins_final.jsonl
( time timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/prof/${R}_bench_line.json 2> gpurun_out/prof/${R}_bench.err ) 2> gpurun_out/prof/${R}_bench_wallclock.txt
cat gpurun_out/prof/${R}_gpu_suite.txt; tail -3 gpurun_out/prof/${R}_bench_wallclock.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/prof/r04_bench_line.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stages_ms", "extras_error")})
for m in ("fp8", "fp8mx"):
    c = d["configs"][m]
    print(m, c.get("value"), c.get("ms_per_step"), (c.get("pipeline") or {}).get("frames_per_s"), (c.get("pipeline") or {}).get("ms_per_frame_in_groups_of_8"), c.get("error"))
print("pipeline", d["pipeline"]["frames_per_s"], d["pipeline"]["ms_per_frame"], d["pipeline"]["ms_per_frame_in_groups_of_8"], d["pipeline"]["stages_ms"])
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
