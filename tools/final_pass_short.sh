# The round-end pass without the counter passes and the sharded run (kernels unchanged since tools/final_pass.sh ran: their summaries stay)
# kernel stats of the bench command on three streams and on one, PMC passes folded into profiles/r03_pmc_summary.json).
#   gpurun --timeout 2700 -- 'bash tools/final_pass.sh'      then copy gpurun_out/prof/* into profiles/
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=r03
mkdir -p gpurun_out/prof; rm -f gpurun_out/margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/prof/${R}_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/prof/${R}_gpu_suite.txt
cp gpurun_out/margins.jsonl gpurun_out/prof/${R}_parity_margins_final.jsonl
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/prof/${R}_bench_line.json 2> gpurun_out/prof/${R}_bench.err
timeout 400 python bench.py --config fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-pipeline > gpurun_out/prof/${R}_bench_fp8_line.json 2>> gpurun_out/prof/${R}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > /dev/null 2>&1
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_bench_kernel_stats.csv
S6D_BENCH_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > /dev/null 2>&1
cp $(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${R}_bench_serial_kernel_stats.csv
timeout 300 python tools/pem_ops_profile.py 32 2>/dev/null | grep -v "Warning\|warn" > gpurun_out/prof/${R}_pem_ops.txt
cat gpurun_out/prof/${R}_gpu_suite.txt
