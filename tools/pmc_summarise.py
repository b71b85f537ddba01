"""Fold rocprofv3 --pmc counter_collection CSVs into profiles/r01_pmc_summary.json (bytes per launch).

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B... the guide (MI355X_MICROARCH.md, HBM) gives
hbm_bytes = counter * 1024 and notes that on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced stream:
the read side is doubled here (`fetch_x2`), the raw value is kept alongside."""
import csv
import glob
import json
import sys
from collections import defaultdict

# Template instances that tools/pmc_kernels.py launches at TWO shapes, in blocks of RUN launches (bench._event_ms: 1 + 10) that
# alternate proj, lin2: their rows are split by dispatch order so that each shape gets its own traffic figure.
RUN = 11
SPLIT = {"gemm_bf16_kernel<2, true>": ("proj", "lin2"), "gemm4_bf16_kernel<2, true>": ("proj", "lin2")}
out = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        seen = defaultdict(int)
        for r in rows:
            k = r["Kernel_Name"]
            base = k.split("s6d::")[1].split("(")[0] if "s6d::" in k else None
            if base in SPLIT:
                i = seen[(k, r["Counter_Name"])]
                seen[(k, r["Counter_Name"])] += 1
                k = k.replace(base, base + " [" + SPLIT[base][(i // RUN) % len(SPLIT[base])] + "]")
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, c in out.items():
    if "s6d::" not in k:
        continue
    name = k.split("s6d::")[1].split("(")[0]
    fetch = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    write = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    summary[name] = {"launches": len(c.get("FETCH_SIZE", [])), "FETCH_SIZE_raw_per_launch": fetch,
                     "WRITE_SIZE_raw_per_launch": write,
                     "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                     "note": "FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads); counters * 1024 B"}
json.dump(summary, open(sys.argv[1], "w"), indent=1)
print(json.dumps(summary, indent=1))
