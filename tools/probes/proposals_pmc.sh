#!/bin/bash
# PMC passes over the proposals stage's kernels (tools/probes/proposals_trace.py): bash tools/probes/proposals_pmc.sh
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pp_$tag -o g -- python $R/tools/probes/proposals_trace.py 3 > /dev/null 2>&1
  python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"/tmp/pp_{tag}/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file for", tag); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if not any(s in k for s in ("tok2img_raw", "img2tok", "upscale_heads", "mask_post", "samtok")): continue
    k = k.split("(")[0][-36:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k, "launches", len(n[k]), {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
done
