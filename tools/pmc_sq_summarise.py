"""Fold the SQ / GRBM counter passes of tools/final_pass.sh into one JSON: per kernel template instance the per-launch averages and the
derived matrix-pipe utilisation.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16),
GRBM_GUI_ACTIVE the busy cycles summed over the 8 XCDs (MI355X_MICROARCH.md): mfma_util = busy / (1024 SIMDs x kernel cycles)."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, c in out.items():
    if "s6d::" not in k:
        continue
    name = k.split("s6d::")[1].split("(")[0]
    row = {n: sum(v) / len(v) for n, v in c.items()}
    row["launches"] = max(len(v) for v in c.values())
    if row.get("GRBM_GUI_ACTIVE") and row.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = row["GRBM_GUI_ACTIVE"] / 8.0
        row["kernel_cycles_per_xcd"] = cyc
        row["mfma_util"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
    summary[name] = row
json.dump(summary, open(sys.argv[1], "w"), indent=1)
print(json.dumps(summary, indent=1)[:4000])
