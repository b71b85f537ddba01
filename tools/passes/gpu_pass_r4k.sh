# round 4: the MX lin1 -> lin2 pair for the DINOv2 MLP (row counts that are not multiples of 256)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/margins.jsonl
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -k "vit_l14 or mx" 2>&1 | tail -5
grep vit_l14 gpurun_out/margins.jsonl
S6D_DINO_GEMM=fp8 timeout 300 python tools/dinov2_time.py 128 2>&1 | grep -v Warn | tail -1
S6D_DINO_GEMM=fp8mx timeout 300 python tools/dinov2_time.py 128 2>&1 | grep -v Warn | tail -1
S6D_DINO_GEMM=fp8mx timeout 300 python tools/dinov2_time.py 255 255 2>&1 | grep -v Warn | tail -1
S6D_DINO_GEMM=fp8 timeout 300 python tools/dinov2_time.py 255 255 2>&1 | grep -v Warn | tail -1
timeout 300 python tools/dinov2_time.py 255 255 2>&1 | grep -v Warn | tail -1
