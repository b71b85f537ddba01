# Round 3, pass g: library ops left in the PEM stage by call site; bench with the whole-frame pipeline block
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 300 python tools/pem_ops_profile.py 32 > $O/1_pem_ops.txt 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/2_bench.json 2> $O/2.err
grep -v amdgpu.ids $O/1_pem_ops.txt | head -50
python -c "
import json; d=json.loads(open('$O/2_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_ms'), d.get('pem_vit_bf16'), d.get('extras_error')); print(json.dumps(d.get('pipeline'), indent=0)[:1500])"
