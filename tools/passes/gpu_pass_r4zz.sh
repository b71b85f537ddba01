# round 4, last seconds of the GPU budget: the whole-frame kernel statistics AT HEAD (after the decoder's cast-cached Linear layers)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
cat > /tmp/fd.py <<'PY'
import json, os, sys, torch
os.environ.setdefault("S6D_PEM_VIT_DTYPE", "fp16")           # as bench.py runs the whole-frame block
sys.path.insert(0, "tools")
import frame_demo
d = frame_demo.measure(torch.device("cuda", 0))
d.pop("_built", None)
print(json.dumps(d))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fd -o fd -- python /tmp/fd.py > gpurun_out/prof/r04_frame_demo.txt 2>/dev/null
cp $(find /tmp/prof_fd -name "*kernel_stats.csv" | head -1) gpurun_out/prof/r04_frame_demo_kernel_stats.csv
tail -1 gpurun_out/prof/r04_frame_demo.txt | cut -c1-700
