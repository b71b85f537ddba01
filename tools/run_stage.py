"""Run one stage of the hot path a few times (for rocprofv3 --kernel-trace --stats)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "sam"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bench.benched_policy()
hp = bench.HotPath(torch.device("cuda", 0), frames, min(frames, bench.SAM_CHUNK))
fn = {"sam": hp.sam_stage, "ism": hp.ism_stage, "pem": hp.pem_stage, "step": hp.step}[stage]
fn()
torch.cuda.synchronize()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done", stage, frames, reps)
