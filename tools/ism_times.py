"""Per-call wall times inside one ISM frame scoring (diagnostic; run on the GPU box).  usage: ism_times.py [P O T]
(default 128 1 42 = the benched LM-O shape; 256 30 42 = BASELINE configs[3], T-LESS: 30 objects, many-instance frames)."""
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from sam6d_amd.ism.scoring import FrameScorer  # noqa: E402
from sam6d_amd.utils import synth  # noqa: E402

P, O, Tn = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 1, 42)
print(f"P = {P} proposals, O = {O} objects, T = {Tn} templates", flush=True)
inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth.ism_inputs(P=P, O=O, T=Tn, seed=11).items()}
fs = FrameScorer(inp["ref_cls"], inp["ref_patch"], inp["poses"], inp["pointcloud"])


def T(name, fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms", flush=True)
    return out


sel, pobj, sem, bt = T("compute_semantic_score", lambda: fs.compute_semantic_score(inp["qry_cls"]))
print("selected", len(sel))
qp = T("qry_patch[sel]", lambda: inp["qry_patch"][sel])
appe, ref = T("compute_appearance_score", lambda: fs.compute_appearance_score(bt, pobj, qp))
masks = T("masks[sel]", lambda: inp["masks"][sel])
batch = dict(depth=[inp["depth"]], cam_intrinsic=[inp["K"]], depth_scale=1.0)
uv = T("project_template_to_image", lambda: fs.project_template_to_image(bt, pobj, batch, masks))
T("  Calculate_the_query_translation", lambda: fs.Calculate_the_query_translation(masks, inp["depth"], inp["K"], 1.0))
boxes = inp["boxes"][sel]
T("compute_geometric_score", lambda: fs.compute_geometric_score(uv, SimpleNamespace(boxes=boxes), qp, ref, 0.5))
T("whole score()", lambda: fs.score(inp["qry_cls"], inp["qry_patch"], inp["masks"], inp["boxes"], inp["depth"], inp["K"]))
