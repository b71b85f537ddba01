"""Launch the hand-written kernels at bench shapes a few times (target of the rocprofv3 --pmc passes), the fp8 kernels of
BASELINE configs[4] included."""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for _ in range(2):
    bench.kernel_rooflines(dev, bench.SAM_CHUNK, 32)
os.environ["S6D_SAM_GEMM"] = "fp8"; __import__("sam6d_amd.policy").policy.reload()
bench.kernel_rooflines(dev, bench.SAM_CHUNK, 32)
os.environ["S6D_SAM_GEMM"] = "fp8mx"; __import__("sam6d_amd.policy").policy.reload()                     # + lin1 with the MX output, lin2 with MX activations (round 4)
bench.kernel_rooflines(dev, bench.SAM_CHUNK, 32)
torch.cuda.synchronize()
print("done")
