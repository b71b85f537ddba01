#!/bin/bash
# Same-box A/B of the decoder kernels' variants: library builds with the given macro at 1 / 0, the proposals trace under each, twice.
#   bash tools/probes/samdec_lds_ab.sh [S6D_SD_LDSFIX | S6D_SD_CHANPERM]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
M=${1:-S6D_SD_LDSFIX}
VALS=${2:-"1 0"}
for rep in 1 2; do
  for v in $VALS; do
    S6D_EXTRA_HIPCC_FLAGS="-D$M=$v" python -c "from sam6d_amd import _lib; _lib.build()" > /dev/null 2>&1
    echo "== $M=$v (round $rep)"
    S6D_EXTRA_HIPCC_FLAGS="-D$M=$v" bash tools/probes/proposals_trace.sh 2>&1 | grep "proposals stage\|tok2img_raw\|img2tok\|upscale_heads" | cut -c1-110
  done
done
