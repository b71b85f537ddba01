"""Only the segmentor stage (prompt encoder + mask decoder + mask post-processing + NMS for 1024 prompts) of tools/frame_demo.py's
frame, N times -- the target of a `rocprofv3 --kernel-trace --stats` pass that lists what the proposals stage launches:
python tools/probes/proposals_trace.py [N]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import frame_demo  # noqa: E402

dev = torch.device("cuda", 0)
pipe, args = frame_demo.build(dev)
img = args[0]
with torch.no_grad():
    emb = pipe._embed([img])
    for _ in range(3):
        pipe._segment(emb, img)
    torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    t = time.perf_counter()
    for _ in range(n):
        pipe._segment(emb, img)
    torch.cuda.synchronize()
print(f"proposals stage: {(time.perf_counter() - t) * 1e3 / n:.2f} ms per frame over {n} frames")
