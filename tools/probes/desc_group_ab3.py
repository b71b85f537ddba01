"""Stage times of FramePipeline.run_group (synchronised after every stage) with and without the cross-frame descriptor batches."""
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import frame_demo  # noqa: E402

dev = torch.device("cuda", 0)
pipe, args = frame_demo.build(dev, 1024, 10, sync_stages=True)
g_args = [args] * 8
for mode in ("1", "0", "1", "0"):
    os.environ["S6D_DESC_GROUP"] = mode
    pipe.run_group(g_args)
    pipe.times.clear()
    pipe.run_group(g_args)
    print(mode, {k: round(v, 1) for k, v in pipe.times.items()}, flush=True)
