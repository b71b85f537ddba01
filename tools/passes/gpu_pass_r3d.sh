# Round 3, pass d: fp8 numerics diagnosis + the rest of the GPU suite after the GEMM / LayerNorm changes
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 300 python tools/probes/fp8_gemm_diag.py > $O/1_fp8_diag.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/2_tests.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl 2>/dev/null
grep -v amdgpu.ids $O/1_fp8_diag.txt; cat $O/2_tests.txt; cat $O/margins.jsonl
