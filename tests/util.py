"""Shared helpers for the test-suite (golden fixtures, seeded weights)."""
import ast
import os

import numpy as np

from sam6d_amd.utils import seeded

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def shapes_from_golden(g, keys="state_keys", shapes="state_shapes"):
    return {str(k): ast.literal_eval(str(s)) for k, s in zip(g[keys], g[shapes])}


def pem_weights(seed=1):
    """Flat {reference state_dict key: tensor}; constructor-computed buffers added by hand."""
    g = golden("pem_b2.npz")
    W = seeded.seeded_state(shapes_from_golden(g), seed)
    return W


def digest(t, stride):
    t = t.detach().double().reshape(-1).cpu()
    return np.array([t.sum().item(), t.abs().sum().item()]), t[::stride].float().numpy()


def assert_digest_close(t, gsum, gsmp, stride, rtol, atol, what=""):
    s, smp = digest(t, stride)
    np.testing.assert_allclose(smp, gsmp, rtol=rtol, atol=atol, err_msg=f"{what}: strided sample")
    # abs-sum compared relatively (sum itself may cancel)
    assert abs(s[1] - gsum[1]) <= rtol * abs(gsum[1]) + atol * t.numel(), f"{what}: abs-sum {s[1]} vs {gsum[1]}"


def dinov2_shapes(cfg):
    """state_dict surface of DinoVisionTransformer (vision_transformer.py:107-170, layers/block.py:53-78)."""
    D, n = cfg["dim"], cfg["img_size"] // cfg["patch"]
    s = {"cls_token": (1, 1, D), "pos_embed": (1, n * n + 1, D), "mask_token": (1, D),
         "patch_embed.proj.weight": (D, 3, cfg["patch"], cfg["patch"]), "patch_embed.proj.bias": (D,),
         "norm.weight": (D,), "norm.bias": (D,)}
    for i in range(cfg["depth"]):
        p = f"blocks.{i}."
        s.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "norm2.weight": (D,), p + "norm2.bias": (D,),
                  p + "attn.qkv.weight": (3 * D, D), p + "attn.qkv.bias": (3 * D,), p + "attn.proj.weight": (D, D),
                  p + "attn.proj.bias": (D,), p + "ls1.gamma": (D,), p + "ls2.gamma": (D,),
                  p + "mlp.fc1.weight": (4 * D, D), p + "mlp.fc1.bias": (4 * D,), p + "mlp.fc2.weight": (D, 4 * D),
                  p + "mlp.fc2.bias": (D,)})
    return s


def record_margin(test, **values):
    """Append the measured distance of a parity assertion from its bound to gpurun_out/margins.jsonl (merged back from the GPU box;
    the copy that is judged lives under profiles/).  Never fails a test."""
    import json
    try:
        d = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "margins.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if not isinstance(v, (list, str)) else v) for k, v in values.items()})) + "\n")
    except Exception:
        pass


def frame_inputs(c):
    """The reference's Data/Example frame (as frozen in tests/golden/example_frame.npz) + P deterministic proposals on it (depth
    windows inside boxes spread over the frame; proposal 0 is the object mask of gen_example) + synthetic descriptors / template
    poses (no checkpoint offline).  Everything a test needs to rebuild the same inputs without /root/reference."""
    g = golden("example_frame.npz")
    depth_mm = g["depth_mm"].astype(np.float32)
    H, W = depth_mm.shape
    boxes_yxyx = [(140, 262, 300, 442), (60, 150, 80, 200), (250, 400, 100, 260), (300, 460, 420, 600), (20, 120, 420, 560),
                  (180, 300, 10, 120), (100, 220, 220, 330), (330, 470, 270, 400), (150, 260, 480, 630), (10, 90, 250, 400)][:c["P"]]
    masks = np.zeros((c["P"], H, W), bool)
    for j, (y1, y2, x1, x2) in enumerate(boxes_yxyx):
        win = depth_mm[y1:y2, x1:x2]
        med = np.median(win[win > 0]) if (win > 0).any() else 0
        masks[j, y1:y2, x1:x2] = (win > med - 90) & (win < med + 90)
    masks[0] = 0
    masks[0, 140:262, 300:442] = (g["depth_mm"][140:262, 300:442] > 900) & (g["depth_mm"][140:262, 300:442] < 1075)
    boxes = np.zeros((c["P"], 4), np.float32)
    for j in range(c["P"]):
        ys, xs = np.nonzero(masks[j])
        boxes[j] = (xs.min(), ys.min(), xs.max(), ys.max())
    from sam6d_amd.utils import synth
    import torch
    d = synth.ism_inputs(P=c["P"], O=c["O"], T=c["T"], C=c["C"], n_patch=c["n_patch"], H=H, W=W, seed=c["seed"])
    return dict(rgb=g["rgb"], depth_mm=torch.from_numpy(depth_mm), K=torch.from_numpy(g["K"]), depth_scale=float(g["depth_scale"]),
                masks=torch.from_numpy(masks).float(), boxes=torch.from_numpy(boxes), qry_cls=d["qry_cls"], qry_patch=d["qry_patch"],
                ref_cls=d["ref_cls"], ref_patch=d["ref_patch"], poses=d["poses"],
                pointcloud=torch.from_numpy(g["dense_po"])[None], model=g["model"], dense_po=g["dense_po"], radius=float(g["radius"]))


