# round 4, second GPU call: the new parity tests (frame_e2e, DINOv2 error model, fp16 range guard), the strip-walk mask post-processing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_zz_frame_e2e.py tests/test_gpu_dinov2.py tests/test_gpu_sam_decoder.py "tests/test_gpu_pem.py::test_fp16_extractor_range_guard_reruns_overflowing_instances_in_fp32" -q 2>&1 | tail -40
cat gpurun_out/margins.jsonl | grep -i "e2e\|vit_l14\|error_growth" 
timeout 300 python tools/sam_decoder_time.py 1024 256 2>&1 | grep -v Warn | tail -8
