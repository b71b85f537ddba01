set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/final_gpu_tests.txt
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_v9.json 2> gpurun_out/bench_v9.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/prof/r01_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python tools/pmc_kernels.py > /dev/null 2>&1
python tools/pmc_summarise.py gpurun_out/prof/r01_pmc_summary.json /tmp/pmc_f /tmp/pmc_w > /dev/null 2>&1
cat gpurun_out/final_gpu_tests.txt
tail -c 600 gpurun_out/bench_v9.json
