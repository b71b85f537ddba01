# Round 3, pass h: library ops left in the PEM stage by call site; pipeline test with the graphed PEM; whole-frame timing
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 300 python tools/pem_ops_profile.py 32 > $O/1_pem_ops.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_zz_pipeline.py tests/test_gpu_zz_host_glue.py -q -m gpu 2>&1 | tail -8 > $O/2_tests.txt
timeout 600 python tools/frame_demo.py > $O/3_frame_demo.txt 2>&1
grep -v "amdgpu.ids\|Warning\|warn" $O/1_pem_ops.txt | head -60; cat $O/2_tests.txt; grep -v amdgpu.ids $O/3_frame_demo.txt | tail -25
