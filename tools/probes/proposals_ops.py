"""Which library operators are left in the proposals stage (prompt encoder + mask decoder + post-processing + NMS, 1024 prompts), by
call site: torch.profiler over one eager (not graph-replayed) call.  python tools/probes/proposals_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sam6d_amd import policy  # noqa: E402
policy.set(amg_graph="0")
import frame_demo  # noqa: E402

dev = torch.device("cuda", 0)
pipe, args = frame_demo.build(dev)
img = args[0]
with torch.no_grad():
    emb = pipe._embed([img])
    for _ in range(2):
        pipe._segment(emb, img)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        pipe._segment(emb, img)
        torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=12)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if dt <= 0 or not e.key.startswith("aten::"):
        continue
    site = next((fr.split("/")[-1] for fr in (e.stack or []) if "sam6d_amd" in fr), "?")
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:60], site[:60]))
tot = sum(r[0] for r in rows)
print(f"aten ops, self device time: {tot / 1e3:.2f} ms in {sum(r[1] for r in rows)} calls")
for dt, c, n, s_, site in sorted(rows, reverse=True)[:70]:
    print(f"{dt / 1e3:7.3f} ms {c:4d}x {n:24s} {s_:60s} {site}")
