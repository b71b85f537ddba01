"""Launch the hand-written kernels at bench shapes a few times (target of the rocprofv3 --pmc passes)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for _ in range(2):
    bench.kernel_rooflines(dev, bench.SAM_CHUNK, 32)
torch.cuda.synchronize()
print("done")
