# round 4: closing pass at HEAD -- suite, smoke, the bench line (timed), clean kernel statistics of the step
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r04
mkdir -p gpurun_out/prof; rm -f gpurun_out/margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/prof/${R}_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/prof/${R}_gpu_suite.txt
cp gpurun_out/margins.jsonl gpurun_out/prof/${R}_parity_margins_final.jsonl
cp profiles/${R}_pmc_summary.json gpurun_out/prof/ 2>/dev/null
( time timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/prof/${R}_bench_line.json 2> gpurun_out/prof/${R}_bench.err ) 2> gpurun_out/prof/${R}_bench_wallclock.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2>&1
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_bench_kernel_stats.csv
S6D_BENCH_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o bench -- python bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2>&1
cp $(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_bench_serial_kernel_stats.csv
cat gpurun_out/prof/${R}_gpu_suite.txt gpurun_out/prof/${R}_bench_wallclock.txt
python - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/prof/r04_bench_line.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stages_ms", "extras_error")})
for m in ("fp8", "fp8mx"):
    c = d["configs"][m]
    print(m, c.get("value"), c.get("ms_per_step"), c.get("pipeline", {}).get("frames_per_s"), c.get("error"))
print("pipeline", d["pipeline"]["frames_per_s"], d["pipeline"]["ms_per_frame_in_groups_of_8"], d["pipeline"]["stages_ms"])
rows = list(csv.DictReader(open("gpurun_out/prof/r04_bench_serial_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
own = sum(float(r["TotalDurationNs"]) for r in rows if "s6d" in r["Name"])
print("kernels of this library: %.1f %% of the step's GPU time; library kernels: %.1f %%" % (100 * own / tot, 100 * (1 - own / tot)))
for r in rows:
    if "s6d" not in r["Name"] and float(r["Percentage"]) > 0.15:
        print("   ", r["Name"][:110], r["Calls"], r["Percentage"])
PY
