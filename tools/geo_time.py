"""geo_embed_kernel timing at the PEM shape (B = 32 clouds of 197 points)."""
import sys

import torch

sys.path.insert(0, ".")
from sam6d_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
B, N = 32, 197
idx4 = (torch.rand(B, N, N, 4, generator=g) * 10).cuda()
Wd, Wa = (torch.randn(256, 256, generator=g) / 16).cuda(), (torch.randn(256, 256, generator=g) / 16).cuda()
bd, ba = torch.randn(256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
div = torch.exp(torch.arange(0, 256, 2).float() * (-9.210340371976184 / 256)).cuda()
for _ in range(2):
    ops.geo_embedding(idx4, Wd, bd, Wa, ba, div)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.geo_embedding(idx4, Wd, bd, Wa, ba, div)
e1.record()
torch.cuda.synchronize()
print("geo_embed ms", e0.elapsed_time(e1) / 5)
