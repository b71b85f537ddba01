cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 900 python tools/probes/group_exact.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5h/group_exact.txt
cat gpurun_out/r5h/group_exact.txt
