"""Which library (ATen / hipBLASLt) operators are left in the PEM stage, by call site: torch.profiler over one pem_stage call
(run on the GPU box).  Prints the top device-time operators with input shapes and the innermost sam6d_amd frame."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hp = bench.HotPath(torch.device("cuda", 0), B, min(B, 16))
with torch.no_grad():
    hp.pem_stage()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        hp.pem_stage()
        torch.cuda.synchronize()
rows = []
for e in prof.events():
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if not e.name.startswith("aten::") or dt <= 0 or e.cpu_children:
        continue
    site = next((f"{fr.split('/')[-1]}" for fr in (e.stack or []) if "sam6d_amd" in fr or "bench.py" in fr), "?")
    rows.append((e.name, str(e.input_shapes)[:70], site[:60], dt))
agg = {}
for n, s, site, dt in rows:
    k = (n, s, site)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += dt
tot = sum(v[1] for v in agg.values())
print(f"leaf aten ops: {tot / 1e3:.2f} ms of device time")
for (n, s, site), (c, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{dt / 1e3:7.3f} ms {c:4d}x {n:28s} {s:70s} {site}")
