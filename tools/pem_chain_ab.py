"""PEM stage time with and without the fused post-attention chain (csrc/s6d_pchain.hip), same process, alternating rounds:
python tools/pem_chain_ab.py [instances ...]   -> gpurun_out/pem_chain_ab.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sam6d_amd import policy  # noqa: E402


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
    rows = {"chain": [], "three_launches": []}
    for _ in range(3):
        rows["chain"].append(round(ms(hp.pem_stage), 3))
        with policy.use(disable_fused="attn_output_chain"):
            rows["three_launches"].append(round(ms(hp.pem_stage), 3))
    out[f"instances_{B}"] = rows
    print(B, rows, flush=True)
    del hp
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pem_chain_ab.json"), "w"), indent=1)
