# Round 3, pass a: the tightened parity tests (a9 bit-exact, B=16 attention, PEM bars, ViT-H error model) + margins
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 900 python -m pytest tests/test_gpu_ism.py tests/test_gpu_attn.py tests/test_gpu_pem.py tests/test_gpu_sam.py -q -m gpu 2>&1 | tail -40 > $O/1_tests.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl
cat $O/1_tests.txt; cat $O/margins.jsonl
