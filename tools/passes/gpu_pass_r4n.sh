# round 4: raw token->image attention in the mask decoder (A/B against the round-3 k/v form) + sequence attention with two workgroups per CU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sam_decoder.py tests/test_gpu_zz_frame_e2e.py tests/test_gpu_dinov2.py tests/test_gpu_zz_pipeline.py -x -q 2>&1 | tail -5
timeout 600 python tools/frame_demo.py 2>&1 | grep -v Warn | tail -4
S6D_SAMDEC_T2I=kv timeout 600 python tools/frame_demo.py 2>&1 | grep -v Warn | tail -1
