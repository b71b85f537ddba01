"""Where does FramePipeline leave the pixels-to-pose golden?  Stage-by-stage comparison on the device (round 5 diagnostic)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for k in ("S6D_SAM_DECODER_DTYPE", "S6D_SAM_DTYPE", "S6D_DINO_DTYPE", "S6D_PEM_VIT_DTYPE"):
    os.environ[k] = "fp32"
from tests import test_gpu_zz_pipeline_e2e as T  # noqa: E402
from tests import util  # noqa: E402
from sam6d_amd.pem import preprocess as pem_pre  # noqa: E402

flow = sys.argv[1] if len(sys.argv) > 1 else "bop"
g, gp, c, pc = T._goldens()
pipe, frame, (dense_po, dense_fo), pin = T.build_pipeline(g, gp, c, pc, flow, bf16=False)
image, depth, K, keys, ru = frame
want = T._golden_flow(gp, flow, pc)
smp = dense_fo.detach().double().reshape(-1).cpu()[::101].float().numpy()
d = np.abs(smp - gp["dense_fo_smp"])
print("template features vs the reference's (every 101st element): max |d|", float(d.max()), "rms d", float(np.sqrt((d ** 2).mean())),
      "rms value", float(np.sqrt((gp["dense_fo_smp"] ** 2).mean())))
with torch.no_grad():
    emb = pipe._embed([image])
    det = pipe._detect_group(emb, [frame])[0]
    print("detections", len(det), "scores", [round(float(s), 4) for s in det.scores])
    radius = pipe.radius.to(det.object_ids.device)[det.object_ids.long()]
    obs = pem_pre.observed_inputs(image, depth, K, det.masks, radius, keys[: det.masks.shape[0]])
    print("kept", obs["kept"].tolist())
    p = flow + "_"
    M = obs["pts"].shape[0]
    for i in range(M):
        print(i, "pts equal", bool(np.array_equal(obs["pts"][i].cpu().numpy(), gp[p + "pts"][i])),
              "max|dpts|", float(np.abs(obs["pts"][i].cpu().numpy() - gp[p + "pts"][i]).max()),
              "rgb_choose equal", bool(np.array_equal(obs["rgb_choose"][i].cpu().numpy(), gp[p + "rgb_choose"][i])),
              "bbox", obs["bbox"][i].tolist(), gp[p + "bbox"][i].tolist())
    util.assert_digest_close(obs["rgb"].cpu(), gp[p + "rgb_sum"], gp[p + "rgb_smp"], 4099, 1e-6, 1e-6, "rgb crops")
    print("rgb crops digest ok")
    oid = det.object_ids[obs["kept"]].long()
    ep = dict(pts=obs["pts"], rgb=obs["rgb"], rgb_choose=obs["rgb_choose"], model=pipe.tpl["model"][oid].contiguous(),
              dense_po=pipe.tpl["dense_po"][oid].contiguous(), dense_fo=pipe.tpl["dense_fo"][oid].contiguous(), coarse_rand_u=ru[:M])

    def cmp(tag, out):
        for k in ("init_R", "init_t", "pred_R", "pred_t"):
            a, b = out[k].cpu().numpy(), gp[p + k]
            d = np.abs(a - b).reshape(M, -1).max(1)
            print(f"  {tag:14s} {k:7s}", " ".join(f"{x:.1e}" for x in d))
    cmp("eager", pipe.pem(dict(ep)))
    cmp("graph", {**pipe._pem_forward(dict(ep)), "init_R": torch.from_numpy(gp[p + "init_R"]), "init_t": torch.from_numpy(gp[p + "init_t"])})
    ep2 = dict(ep)
    gq = torch.Generator().manual_seed(5)
    for k in ("rgb", "dense_fo"):
        ep2[k] = ep[k] * (1 + 1e-6 * torch.randn(ep[k].shape, generator=gq).to(ep[k].device))
    cmp("eager+1e-6", pipe.pem(dict(ep2)))
    one = {k: v[1:2].contiguous() for k, v in ep.items()}
    o1 = pipe.pem(one)
    print("  instance 1 alone: |pred_R - golden|", float(np.abs(o1["pred_R"].cpu().numpy() - gp[p + "pred_R"][1:2]).max()),
          "|init_R - golden|", float(np.abs(o1["init_R"].cpu().numpy() - gp[p + "init_R"][1:2]).max()))
# the oracle's Net (CPU restatement, test infrastructure) on the SAME inputs: is the golden reproduced outside the reference?
from oracle import pem as opem  # noqa: E402
from sam6d_amd.utils import seeded  # noqa: E402
W = {k: v.detach().cpu().clone() for k, v in pipe.pem.state_dict().items()}
# ---- the oracle's pre-processing on the same detections: per-instance differences of the colour crops and pixel indices
from oracle import pem_pre as opre  # noqa: E402
fi = pin["fi"]
depth_np = fi["depth_mm"].numpy() * np.float32(fi["depth_scale"]) / np.float32(1000.0)
oobs = opre.preprocess_frame(fi["rgb"], depth_np, fi["K"].numpy(), det.masks.cpu().numpy(), radius.cpu().numpy(), keys=keys[: det.masks.shape[0]].cpu().numpy())
for i in range(M):
    dr = np.abs(obs["rgb"][i].cpu().numpy() - oobs["rgb"][i])
    dc = obs["rgb_choose"][i].cpu().numpy() != oobs["rgb_choose"][i]
    print(i, "rgb crop: max diff", float(dr.max()), "pixels differing", int((dr.max(0) > 0).sum()), "| rgb_choose differing", int(dc.sum()),
          "first", [(int(a), int(b)) for a, b in zip(obs["rgb_choose"][i].cpu().numpy()[dc][:4], oobs["rgb_choose"][i][dc][:4])])
sub = [5, 6]
epc = {k: v[sub].cpu() for k, v in ep.items() if k != "coarse_rand_u"}
with torch.no_grad():
    oo = opem.net_forward(W, epc, ru[sub].cpu())
for k in ("init_R", "pred_R"):
    print("  oracle Net rows", sub, "on the product's inputs", k, np.abs(oo[k].numpy() - gp[p + k][sub]).reshape(2, -1).max(1))
