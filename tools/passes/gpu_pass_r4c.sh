# round 4, third GPU call: the strip-pair sequence attention (A/B against the window kernel), its device tests, the range guard test
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
S6D_SEQ_ATTN_IMPL=1 timeout 120 python tools/seq_attn_time.py 2>&1 | grep impl
timeout 120 python tools/seq_attn_time.py 2>&1 | grep impl
timeout 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_dinov2.py tests/test_gpu_pem.py -q -x 2>&1 | tail -5
timeout 300 python tools/dinov2_time.py 128 2>&1 | grep -v Warn | tail -4
