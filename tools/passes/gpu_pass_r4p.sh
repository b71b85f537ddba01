# round 4: W_p folded into the RPE layer's projection -- parity + A/B of the PEM stage in one process each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pose.py tests/test_gpu_pem.py tests/test_gpu_plin.py -x -q 2>&1 | tail -3
cat > /tmp/pem_ab.py <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
os.environ["S6D_PEM_VIT_DTYPE"] = "fp16"
import bench
hp = bench.HotPath(torch.device("cuda:0"), 32, 16)
for fold in ("1", "0", "1", "0"):
    os.environ["S6D_RPE_FOLD"] = fold
    print("S6D_RPE_FOLD=" + fold, "PEM stage ms per 32 instances:", round(bench.stage_ms(hp.pem_stage, 5), 3), flush=True)
PY
python /tmp/pem_ab.py 2>&1 | grep FOLD
