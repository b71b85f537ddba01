cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
V=tools/gemm_variants
timeout 600 python tools/probes/gemm_ab.py base=$V/libgemm_base.so ant=$V/libgemm_ant.so bnt=$V/libgemm_bnt.so abnt=$V/libgemm_abnt.so 2>&1 | grep -v amdgpu.ids
