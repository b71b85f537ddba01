# Round-end measurement pass (tests + smoke, bench line with cpu_baseline + configs.fp8 + the whole-frame pipeline block, rocprofv3 kernel
# stats of the bench command on three streams and on one and of the whole-frame demo, PMC passes folded into profiles/r06_*.json).
#   gpurun --timeout 3000 -- 'bash tools/final_pass.sh'      then copy gpurun_out/prof/* into profiles/
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r06
mkdir -p gpurun_out/prof; rm -f gpurun_out/margins.jsonl
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/prof/${R}_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/prof/${R}_gpu_suite.txt
cp gpurun_out/margins.jsonl gpurun_out/prof/${R}_parity_margins_final.jsonl
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python tools/pmc_kernels.py > /dev/null 2>&1
python tools/pmc_summarise.py gpurun_out/prof/${R}_pmc_summary.json /tmp/pmc_f /tmp/pmc_w > /dev/null 2>&1
cp gpurun_out/prof/${R}_pmc_summary.json profiles/${R}_pmc_summary.json   # the bench line below reads its roofline.traffic from it
( time timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/prof/${R}_bench_line.json 2> gpurun_out/prof/${R}_bench.err ) 2> gpurun_out/prof/${R}_bench_wallclock.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 8 --warmup 1 --no-extras > /dev/null 2>&1
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_bench_kernel_stats.csv
S6D_BENCH_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o bench -- python bench.py --steps 8 --warmup 1 --no-extras > /dev/null 2>&1
cp $(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_bench_serial_kernel_stats.csv
cat > /tmp/fd.py <<'PY'
import json, os, sys, torch
os.environ.setdefault("S6D_PEM_VIT_DTYPE", "fp16")           # as bench.py runs the whole-frame block
sys.path.insert(0, "tools")
import frame_demo
d = frame_demo.measure(torch.device("cuda", 0))
d.pop("_built", None)
print(json.dumps(d))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fd -o fd -- python /tmp/fd.py > gpurun_out/prof/${R}_frame_demo.txt 2>/dev/null
cp $(find /tmp/prof_fd -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_frame_demo_kernel_stats.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_s -o s -- python tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_g -o g -- python tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_i -o i -- python tools/pmc_kernels.py > /dev/null 2>&1
python tools/pmc_sq_summarise.py gpurun_out/prof/${R}_sq_summary.json /tmp/pmc_s /tmp/pmc_g /tmp/pmc_i > /dev/null 2>&1
timeout 300 python tools/pem_ops_profile.py 32 2>/dev/null | grep -v "Warning\|warn" > gpurun_out/prof/${R}_pem_ops.txt
timeout 600 python tools/run_sharded.py --frames 16 --group 8 --out gpurun_out/prof/${R}_sharded_world1.csv --fixed-time 0 2>/dev/null | tail -1 > gpurun_out/prof/${R}_sharded_world1.json
cat gpurun_out/prof/${R}_gpu_suite.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/prof/r06_bench_line.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stages_ms", "roofline", "extras_error")})
print("pipeline", d.get("pipeline"))
print("cpu", d.get("cpu_baseline"))
for m in ("fp8", "fp8mx"):
    c = d.get("configs", {}).get(m, {})
    print(m, c.get("value"), c.get("ms_per_step"), (c.get("pipeline") or {}).get("frames_per_s"), c.get("error"))
import csv
rows = list(csv.DictReader(open("gpurun_out/prof/r06_bench_serial_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
own = sum(float(r["TotalDurationNs"]) for r in rows if "s6d" in r["Name"])
print("kernels of this library: %.1f %% of the traced GPU time; library kernels: %.1f %%" % (100 * own / tot, 100 * (1 - own / tot)))
for k in d.get("kernels", []):
    print(k["kernel"][:100], k["avg_ms"], k["frac"], k.get("pmc_key"))
PY
