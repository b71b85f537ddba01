"""Does hipGraph capture of a stage pay?  Eager vs replayed timing of the PEM and SAM stages (run on the GPU box)."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

bench._use_tuned_library_gemms()
bench.benched_policy()
hp = bench.HotPath(torch.device("cuda", 0), 32, bench.SAM_CHUNK)
for name, fn in (("pem", hp.pem_stage), ("sam", hp.sam_stage)):
    eager = bench.stage_ms(fn, 3)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = fn()
        rep = bench.stage_ms(g.replay, 3)
        print(f"{name}: eager {eager:.2f} ms, graph replay {rep:.2f} ms", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
