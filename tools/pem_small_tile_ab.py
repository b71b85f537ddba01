"""PEM stage time with and without the 256 x 128 tile form for its under-filled ViT-B GEMMs (s6d_set_gemm_small_tile), same process,
alternating rounds:  python tools/pem_small_tile_ab.py [instances ...]  -> gpurun_out/pem_small_tile_ab.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sam6d_amd import ops  # noqa: E402


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


out = {}
bench.benched_policy()
for B in [int(a) for a in sys.argv[1:]] or [32, 10]:
    hp = bench.HotPath(torch.device("cuda", 0), B, min(B, 16))
    rows = {"small_tile": [], "tiles_256": []}
    for _ in range(3):
        ops.set_gemm_small_tile(True)
        rows["small_tile"].append(round(ms(hp.pem_stage), 3))
        ops.set_gemm_small_tile(False)
        rows["tiles_256"].append(round(ms(hp.pem_stage), 3))
    ops.set_gemm_small_tile(True)
    out[f"instances_{B}"] = rows
    print(B, rows, flush=True)
    del hp
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pem_small_tile_ab.json"), "w"), indent=1)
