#!/bin/bash
# kernel list of the proposals stage: bash tools/probes/proposals_trace.sh  -> gpurun_out/proposals_kernel_stats.csv + a summary
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o p -- python $R/tools/probes/proposals_trace.py 10 2>&1 | grep "proposals stage"
f=$(find /tmp/ptr -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/proposals_kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = 13  # 3 warm-up + 10 timed frames
lib = [r for r in rows if "s6d" not in r["Name"]]
print(f"GPU time per frame {tot / n / 1e6:.2f} ms; s6d share {100 * (1 - sum(float(r['TotalDurationNs']) for r in lib) / tot):.1f} %; "
      f"launches per frame: s6d {sum(int(r['Calls']) for r in rows if 's6d' in r['Name']) / n:.0f}, library {sum(int(r['Calls']) for r in lib) / n:.0f}")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs']) / n / 1e6:7.3f} ms/frame {int(r['Calls']) / n:7.1f} calls/frame {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
PY
