# round 4, fourth GPU call: the one-round sequence attention (A/B against the window kernel), device tests, descriptor stage time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
S6D_SEQ_ATTN_IMPL=1 timeout 120 python tools/seq_attn_time.py 2>&1 | grep impl
timeout 120 python tools/seq_attn_time.py 2>&1 | grep impl
timeout 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_dinov2.py tests/test_gpu_pem.py -q 2>&1 | tail -5
timeout 300 python tools/dinov2_time.py 128 2>&1 | grep -v Warn | tail -2
