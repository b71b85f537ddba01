# First GPU call of the next round: device checks of everything written after the round-1 GPU budget ran out, then the two
# opt-in experiments, A/B against the default bench.  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/next_gpu_pass.sh'
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 1. new host-glue device tests first (short), then the whole suite
timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py tests/test_gpu_zz_pipeline.py -q 2>&1 | tail -5 > gpurun_out/n1_glue.txt
S6D_PEM_SEQ_CENTROID=1 timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py -q -k sequential 2>&1 | tail -5 > gpurun_out/n2_seq_centroid.txt
S6D_PEM_SAMPLER=kernel timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py tests/test_gpu_pem_pre.py -q -s 2>&1 | tail -8 > gpurun_out/n2_sampler_kernel.txt
S6D_PEM_PRE=kernels timeout 600 python -m pytest tests/test_gpu_zz_host_glue.py tests/test_gpu_pem_pre.py tests/test_gpu_zz_pipeline.py -q -s 2>&1 | tail -10 > gpurun_out/n2_pem_pre_kernels.txt
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/n3_gpu_suite.txt
# 2. the segmentor plugin end to end (ViT-H, seeded weights) and the five-model frame chain
timeout 600 python tools/segmentor_demo.py > gpurun_out/n4_segmentor_demo.txt 2>&1
timeout 600 python tools/frame_demo.py > gpurun_out/n4_frame_demo.txt 2>&1
timeout 300 python tools/pem_pre_time.py 64 > gpurun_out/n4_pem_pre_stages.txt 2>&1
# 3. bench A/B: default vs Infinity-Cache-sized MLP row chunks
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/n5_bench_default.json 2> gpurun_out/n5.err
S6D_SAM_MLP_ROWS=16384 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/n6_bench_mlp16384.json 2> gpurun_out/n6.err
S6D_SAM_MLP_ROWS=32768 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/n7_bench_mlp32768.json 2> gpurun_out/n7.err
for f in gpurun_out/n[1-4]*.txt; do echo "== $f"; cat $f | tail -6; done
for f in gpurun_out/n[5-7]*.json; do echo "== $f"; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; done
# 4. where do the attention kernels' wave cycles go?  (SQ counters, 8 slots per pass; counter runs carry --kernel-trace only)
mkdir -p gpurun_out/sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d /tmp/pmc_sq1 -o sq1 -- python tools/pmc_attn.py > gpurun_out/sq/pass1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace --output-format csv -d /tmp/pmc_sq2 -o sq2 -- python tools/pmc_attn.py > gpurun_out/sq/pass2.log 2>&1
for d in /tmp/pmc_sq1 /tmp/pmc_sq2; do find $d -name "*counter_collection.csv" -exec cp {} gpurun_out/sq/$(basename $d).csv \; ; done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/sq/*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print("==", f)
    for k, d in acc.items():
        print(k, {c: round(v / max(n[(k, c)], 1)) for c, v in d.items()})
PY
