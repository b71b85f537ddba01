"""The PEM stage of the benched step at B instances, N times (target of rocprofv3 --kernel-trace --stats): python tools/probes/pem_trace.py B N"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bench.benched_policy()
hp = bench.HotPath(torch.device("cuda", 0), B, min(B, 16))
for _ in range(3):
    hp.pem_stage()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N):
    hp.pem_stage()
torch.cuda.synchronize()
print(f"pem stage at {B} instances: {(time.perf_counter() - t) * 1e3 / N:.2f} ms")
