cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python tools/probes/e2e_diag.py bop > gpurun_out/r5b/e2e_diag_bop.txt 2>&1
tail -45 gpurun_out/r5b/e2e_diag_bop.txt
