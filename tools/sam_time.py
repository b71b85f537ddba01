"""SAM stage timing (optionally under PYTORCH_TUNABLEOP_ENABLED=1).  usage: sam_time.py [frames] [chunk]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if len(sys.argv) > 3 and sys.argv[3] == "tuned":
    bench._use_tuned_library_gemms()
bench.benched_policy()
hp = bench.HotPath(torch.device("cuda", 0), frames, chunk)
t0 = time.time()
hp.sam_stage()
torch.cuda.synchronize()
print("first call s", time.time() - t0)
ms = bench.stage_ms(hp.sam_stage, 3)
print(f"sam ms/{frames} frames (chunk {chunk})", ms, "-> ms/frame", ms / frames)
