"""SAM stage timing (optionally under PYTORCH_TUNABLEOP_ENABLED=1)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

hp = bench.HotPath(torch.device("cuda", 0), 8, 8)
t0 = time.time()
hp.sam_stage()
torch.cuda.synchronize()
print("first call s", time.time() - t0)
print("sam ms/8 frames", bench.stage_ms(hp.sam_stage, 3))
