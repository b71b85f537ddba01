# full GPU suite + bench line with the folded block loops
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O; rm -f gpurun_out/margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/1_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O/1_suite.txt
cp gpurun_out/margins.jsonl $O/margins.jsonl
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/2_bench.json 2> $O/2_bench.err
cat $O/1_suite.txt; tail -c 3000 $O/2_bench.json
