# Round 2, first GPU pass: the new bf16 GEMM (parity, timing against hipBLASLt, profiling variants, counters), the PEM
# pre-processing kernels that had no device run in round 1, and the bench A/B with / without the kernel.
#   gpurun --timeout 1700 -- 'bash tools/gpu_pass_r2a.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -15 > $O/1_gemm_tests.txt
timeout 600 python tools/gemm_time.py full > $O/2_gemm_time.txt 2>&1; cp gpurun_out/gemm_time.json $O/ 2>/dev/null
# skipped-in-round-1 device tests of the pre-processing kernels
S6D_PEM_PRE=kernels S6D_PEM_SEQ_CENTROID=1 S6D_PEM_SAMPLER=kernel timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py tests/test_gpu_pem_pre.py -q -s -m gpu 2>&1 | tail -15 > $O/3_pem_pre_kernels.txt
timeout 300 python tools/pem_pre_time.py 64 > $O/3_pem_pre_stages.txt 2>&1
# bench A/B
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/4_bench_gemm.json 2> $O/4.err
S6D_DISABLE_FUSED=gemm_bf16 timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/5_bench_library.json 2> $O/5.err
# whole GPU suite (SAM / DINOv2 / PEM ViT goldens now run through the kernel)
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/6_gpu_suite.txt
# counters of the GEMM kernel (separate passes, --kernel-trace only)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d /tmp/pmc_g1 -o g1 -- python tools/gemm_time.py pmc > $O/7_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace --output-format csv -d /tmp/pmc_g2 -o g2 -- python tools/gemm_time.py pmc > $O/7_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_g3 -o g3 -- python tools/gemm_time.py pmc > $O/7_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_g4 -o g4 -- python tools/gemm_time.py pmc > $O/7_pmc4.log 2>&1
for d in g1 g2 g3 g4; do find /tmp/pmc_$d -name "*counter_collection.csv" -exec cp {} $O/pmc_$d.csv \; ; done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r2a/pmc_*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print("==", f)
    for k, d in acc.items():
        if "gemm" in k:
            print(k, {c: round(v / max(n[(k, c)], 1)) for c, v in d.items()})
PY
# kernel-trace stats of the bench with the kernel in
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/8_bench_prof.log 2>&1
find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $O/8_bench_kernel_stats.csv \;
for f in $O/1_*.txt $O/3_*.txt $O/6_*.txt; do echo "== $f"; tail -8 $f; done
echo "== gemm_time"; cat $O/2_gemm_time.txt | tail -45
for f in $O/4_bench_gemm.json $O/5_bench_library.json; do echo "== $f"; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; tail -3 ${f%_*}.err 2>/dev/null; done
head -12 $O/8_bench_kernel_stats.csv | cut -c1-160
