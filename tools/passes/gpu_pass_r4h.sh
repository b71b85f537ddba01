# round 4: fp8 qkv / fc1 for the DINOv2 ViT-L (configs[4]): gate test, descriptor stage time bf16 vs fp8
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -k vit_l14 2>&1 | tail -5
grep vit_l14_fp8 gpurun_out/margins.jsonl
timeout 300 python tools/dinov2_time.py 128 2>&1 | grep -v Warn | tail -1
S6D_DINO_GEMM=fp8 timeout 300 python tools/dinov2_time.py 128 2>&1 | grep -v Warn | tail -1
