"""geo_embed2_kernel (round 6) against the two-phase kernel of rounds 3-5, same process, alternating rounds, at 32 and 10 instances:
python tools/geo_embed_ab.py -> gpurun_out/geo_embed_ab.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sam6d_amd import ops  # noqa: E402
from sam6d_amd.pem.layers import GeometricStructureEmbedding  # noqa: E402
from sam6d_amd.pem.pose_estimation_model import default_cfg  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402


def ms(fn, n=20):
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


geo = seeded.load_seeded(GeometricStructureEmbedding(default_cfg().geo_embedding).eval(), 4).cuda()
split = geo._split_weights()
out = {}
for B in (32, 10):
    pts = synth.pem_inputs(B, seed=9, n_pts=197, with_rgb=False)["pts"].cuda() * 5
    d_idx, a_idx = geo.get_embedding_indices(pts)
    idx4 = torch.cat([d_idx.unsqueeze(-1), a_idx], dim=-1).contiguous()
    args = (idx4, geo.proj_d.weight.contiguous(), geo.proj_d.bias, geo.proj_a.weight.contiguous(), geo.proj_a.bias, geo.embedding.div_term.contiguous())
    rows = {"form2_ms": [], "form1_ms": []}
    for _ in range(3):
        for form in (2, 1):
            ops.set_geo_embed_form(form)
            rows[f"form{form}_ms"].append(round(ms(lambda: ops.geo_embedding(*args, split=split)), 4))
    ops.set_geo_embed_form(2)
    a = ops.geo_embedding(*args, split=split)
    ops.set_geo_embed_form(1)
    b = ops.geo_embedding(*args, split=split)
    ops.set_geo_embed_form(1)
    rows["bit_equal"] = bool(torch.equal(a, b))
    NP = idx4.numel() // 4
    rows["mfma_tflops_form2"] = round(NP * 4 * 256 * 256 * 2 * 3 / (min(rows["form2_ms"]) * 1e-3) / 1e12, 1)
    out[f"instances_{B}"] = rows
    print(B, rows, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "geo_embed_ab.json"), "w"), indent=1)

# ---- the PEM stage of the benched step under both forms (same process, alternating)
import bench  # noqa: E402
bench.benched_policy()
for B in (32, 10):
    hp = bench.HotPath(torch.device("cuda", 0), B, min(B, 16))
    rows = {"form2_ms": [], "form1_ms": []}
    for _ in range(3):
        for form in (2, 1):
            ops.set_geo_embed_form(form)
            rows[f"form{form}_ms"].append(round(ms(hp.pem_stage, 10), 3))
    ops.set_geo_embed_form(1)
    out[f"pem_stage_{B}"] = rows
    print("pem stage", B, rows, flush=True)
    del hp
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "geo_embed_ab.json"), "w"), indent=1)
