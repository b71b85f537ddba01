cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
timeout 600 python tools/attn_time.py 16 > gpurun_out/r5m/attn_time.txt 2>&1
grep -v seq257 gpurun_out/r5m/attn_time.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_sam.py -q 2>&1 | tail -3
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_s -o s -- python tools/pmc_attn.py > /dev/null 2>&1
python tools/pmc_sq_summarise.py gpurun_out/r5m/r05_sq_attn_stream.json /tmp/pmc_s 2>&1 | grep -A8 "attn_window16p_kernel<80, true>" | head -12
