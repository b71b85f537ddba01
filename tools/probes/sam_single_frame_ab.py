"""The SAM ViT-H encoder on ONE frame (M = 4096 rows) with and without the 256 x 128 tile kernel for its under-filled GEMMs
(s6d_set_gemm_small_tile: proj / lin2 + residual + row statistics are 80 tiles of 256 x 256 on 256 CUs), same process, alternating:
python tools/probes/sam_single_frame_ab.py  -> gpurun_out/sam_single_frame_ab.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sam6d_amd import ops  # noqa: E402
import frame_demo  # noqa: E402

dev = torch.device("cuda", 0)
pipe, args = frame_demo.build(dev)
img = args[0]


def ms(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {"small_tile": [], "tiles_256": []}
embs = {}
with torch.no_grad():
    for _ in range(3):
        for key, on in (("small_tile", 2), ("tiles_256", 0)):
            ops.set_gemm_small_tile(on)
            out[key].append(round(ms(lambda: pipe._embed([img])), 3))
            embs[key] = pipe._embed([img])
ops.set_gemm_small_tile(True)
out["embedding_bit_equal"] = bool(torch.equal(embs["small_tile"], embs["tiles_256"]))
print(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sam_single_frame_ab.json"), "w"), indent=1)
