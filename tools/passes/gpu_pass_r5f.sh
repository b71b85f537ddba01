cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
for dt in fp16 fp32; do echo "== S6D_PEM_VIT_DTYPE=$dt"; timeout 600 python tools/probes/batch_invariance.py $dt 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5f/batch_invariance.txt
cat gpurun_out/r5f/batch_invariance.txt
