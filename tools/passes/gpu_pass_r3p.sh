# where the residual K tiles' time goes: ablation builds of EPI 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
V=tools/gemm_variants
timeout 600 python tools/probes/gemm_ab.py base=$V/libgemm_base.so rfromA=$V/libgemm_rfromA.so wnotid=$V/libgemm_wnotid.so both=$V/libgemm_both.so alllive=$V/libgemm_alllive.so > $O/0_gemm_ab.txt 2>&1
grep -v amdgpu.ids $O/0_gemm_ab.txt
