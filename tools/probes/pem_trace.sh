#!/bin/bash
# kernel list of the PEM stage at B instances: bash tools/probes/pem_trace.sh [B]
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-10}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pemtr -o p -- python $R/tools/probes/pem_trace.py $B 20 2>&1 | grep "pem stage"
f=$(find /tmp/pemtr -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/pem_kernel_stats_$B.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = 23
lib = [r for r in rows if "s6d" not in r["Name"]]
print(f"GPU time per pass {tot / n / 1e6:.2f} ms; launches per pass: s6d {sum(int(r['Calls']) for r in rows if 's6d' in r['Name']) / n:.0f}, library {sum(int(r['Calls']) for r in lib) / n:.0f} ({sum(float(r['TotalDurationNs']) for r in lib) / n / 1e6:.2f} ms)")
for r in rows[:32]:
    print(f"{float(r['TotalDurationNs']) / n / 1e6:7.3f} ms/pass {int(r['Calls']) / n:6.1f} calls {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
PY
