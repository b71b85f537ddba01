"""Which library (ATen / hipBLASLt) operators are left in the PEM stage, by call site: torch.profiler over one pem_stage call
(run on the GPU box).  Prints the top device-time operators with input shapes and the innermost sam6d_amd frame."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bench.benched_policy()
hp = bench.HotPath(torch.device("cuda", 0), B, min(B, 16))
with torch.no_grad():
    hp.pem_stage()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        hp.pem_stage()
        torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    site = next((fr.split("/")[-1] for fr in (e.stack or []) if "sam6d_amd" in fr or "bench.py" in fr), "?")
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:64], site[:70]))
tot = sum(r[0] for r in rows)
print(f"aten ops, self device time: {tot / 1e3:.2f} ms")
for dt, c, n, s_, site in sorted(rows, reverse=True)[:45]:
    print(f"{dt / 1e3:7.3f} ms {c:4d}x {n:26s} {s_:64s} {site}")
