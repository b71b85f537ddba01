"""Launch only the two SAM attention kernels (target of SQ counter passes)."""
import sys

import torch

sys.path.insert(0, ".")
from sam6d_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
nh, hd, H = 16, 80, 64
qkv = torch.randn(8, H, H, 3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
bias = torch.randn(3 * nh * hd, generator=g).cuda().to(torch.bfloat16)
for ws, S in ((0, 64), (14, 14)):
    rh = torch.randn(2 * S - 1, hd, generator=g).cuda().to(torch.bfloat16)
    rw = torch.randn(2 * S - 1, hd, generator=g).cuda().to(torch.bfloat16)
    for _ in range(3):
        ops.window_attention(qkv, bias, rh, rw, nh, ws, hd ** -0.5)
torch.cuda.synchronize()
