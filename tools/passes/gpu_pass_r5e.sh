cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
timeout 900 python tools/probes/e2e_diag2.py 6 7 2 > gpurun_out/r5e/e2e_diag2.txt 2>&1
tail -12 gpurun_out/r5e/e2e_diag2.txt
