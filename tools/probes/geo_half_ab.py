"""The geometric embedding stored in IEEE half (S6D_PEM_GEO_DTYPE=fp16) against float32: PEM stage time at B = 32 and at 10 instances
(graph replay, as FramePipeline runs it), and the distance of the poses between the two settings on the same inputs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("S6D_PEM_VIT_DTYPE", "fp16")
from sam6d_amd.pem import pose_estimation_model as pm  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402

dev = torch.device("cuda", 0)
net = seeded.load_seeded(pm.Net(pm.default_cfg()).eval(), 1).to(dev)
keys = ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo", "coarse_rand_u")
res = {}
for B in (32, 10):
    pin = synth.pem_inputs(B, seed=1)
    ep = {k: v.to(dev) for k, v in pin.items() if torch.is_tensor(v)}
    ep["coarse_rand_u"] = synth.coarse_uniforms(B, 2).to(dev)
    ep = {k: ep[k] for k in keys}
    for mode in ("fp32", "fp16"):
        os.environ["S6D_PEM_GEO_DTYPE"] = mode
        with torch.no_grad():
            for _ in range(2):
                out = net(dict(ep))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                net(dict(ep))
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                outg = net(dict(ep))
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 5 * 1e3)
        res[(B, mode)] = (min(ts), {k: outg[k].clone() for k in ("pred_R", "pred_t", "pred_pose_score", "init_R")})
        print(f"B={B:3d} geo {mode}: PEM forward (graph replay) {min(ts):7.3f} ms", flush=True)
    a, b = res[(B, "fp32")][1], res[(B, "fp16")][1]
    dR = (a["pred_R"] - b["pred_R"]).flatten(1).norm(dim=1)
    dt = (a["pred_t"] - b["pred_t"]).norm(dim=1) * 1e3
    same_init = (a["init_R"] - b["init_R"]).flatten(1).norm(dim=1) < 1e-4
    print(f"B={B}: half vs float32 embedding: coarse pose unchanged for {int(same_init.sum())}/{B}; on those dR max {dR[same_init].max().item():.2e} "
          f"dt max {dt[same_init].max().item():.2e} mm; all: dR max {dR.max().item():.2e} dt max {dt.max().item():.2e} mm; "
          f"pose score diff max {(a['pred_pose_score'] - b['pred_pose_score']).abs().max().item():.2e}")
