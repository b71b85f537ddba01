cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 600 python tools/attn_time.py 16 > gpurun_out/r5j/attn_time.txt 2>&1
grep -v seq257 gpurun_out/r5j/attn_time.txt | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_sam.py -q 2>&1 | tail -3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_s -o s -- python tools/pmc_attn.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_g -o g -- python tools/pmc_attn.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_i -o i -- python tools/pmc_attn.py > /dev/null 2>&1
python tools/pmc_sq_summarise.py gpurun_out/r5j/r05_sq_attn.json /tmp/pmc_s /tmp/pmc_g /tmp/pmc_i 2>&1 | grep -A16 "attn_window16p_kernel<80, true>\|attn_global64_kernel<80" | cut -c1-200 | head -50
