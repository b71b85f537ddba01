"""Same-box A/B of FramePipeline.run_group with the descriptor ViT batched across the group's frames (S6D_DESC_GROUP=1) and frame by
frame (=0), and of DINOv2 alone at the batch sizes involved."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import frame_demo  # noqa: E402

dev = torch.device("cuda", 0)
pipe, args = frame_demo.build(dev, 1024, 10, sync_stages=False)
g_args = [args] * 8
for mode in ("1", "0", "1", "0"):
    os.environ["S6D_DESC_GROUP"] = mode
    pipe.run_group(g_args)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        pipe.run_group(g_args)
    torch.cuda.synchronize()
    print(f"S6D_DESC_GROUP={mode}: {(time.perf_counter() - t) * 1e3 / 24:.2f} ms per frame in groups of 8", flush=True)
m = pipe.desc.model
for n in (128, 255, 256):
    x = torch.randn(n, 3, 224, 224, device=dev)
    m(x, is_training=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        m(x, is_training=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / 5
    print(f"DINOv2 ViT-L forward, {n} crops: {ms:.2f} ms = {ms / n * 128:.2f} ms per 128 crops", flush=True)
