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




E2E_POSE_CASE = dict(det_score_thresh=0.46, nms_thresh=0.25, min_box_size=0.05, min_mask_size=3e-4, pem_weight_seed=1, key_seed=41,
                     rand_seed=43, tem_seed=44, scene_id=2, frame_id=17, dataset="ycbv", n_keys=32, base_windows=(0, 8, 9))


def e2e_pose_inputs(c=None, frozen=None):
    """Model-independent inputs of the pose half of the pixels-to-pose golden (tests/golden/frame_e2e_pose.npz), shared by the
    generator (oracle/gen_golden.py frame_e2e_pose) and the tests.

    The PEM's three objects are made FROM THE FRAME so that the matching problem is a real one (seeded weights carry no learned
    prior, but equal inputs give equal features: the known-answer construction of SURVEY.md 8c): object o is the surface seen
    through depth window base_windows[o] of the Example frame.  The oracle's pre-processing of that window (its own seeded
    sampling keys) gives 2048 camera-frame points, the 224 x 224 colour crop and the pixel index of every point; with a seeded
    pose (R0, t0 = the cloud's centroid) the object-frame points are (p - t0) R0, and the onboarding pass (get_obj_feats) gets TWO
    template views of them: the same crop, points [0, 1400) and [600, 2048) (the second set moved by 2e-4 of seeded noise so that
    no two points coincide), each with its pixel indices.  ``model`` = the first 1024 object-frame points, radius = max |model|.
    A detection of that window is then an observation of the object at pose (R0, t0), sampled at other pixels: the Net should
    recover it; detections of other windows assigned to the object stay unrelated (and ill-conditioned: see `stable` in the
    golden).  Everything is a pure function of the frozen Example frame and CPU generators."""
    import torch

    from oracle import pem_pre as opre
    c = c or E2E_POSE_CASE
    fi = frame_inputs(dict(P=10, O=1, T=6, C=128, n_patch=64, seed=21))
    rgb = fi["rgb"]
    H, W = rgb.shape[:2]
    gen = torch.Generator().manual_seed(c["tem_seed"])
    depth = fi["depth_mm"].numpy() * np.float32(fi["depth_scale"]) / np.float32(1000.0)
    base = list(c["base_windows"])
    tkeys = torch.rand(len(base), H * W, generator=gen).numpy()
    obs = opre.preprocess_frame(rgb, depth, fi["K"].numpy(), fi["masks"].numpy()[base] > 0, 10.0, keys=tkeys)
    assert obs["kept"].tolist() == list(range(len(base)))
    pts_cam = torch.from_numpy(obs["pts"])                                     # (O,2048,3)
    crop, choose = torch.from_numpy(obs["rgb"]), torch.from_numpy(obs["rgb_choose"])
    if frozen is not None:
        t0, R0 = torch.from_numpy(frozen["gt_t"]), torch.from_numpy(frozen["gt_R"])
        obj_pts, v1 = torch.from_numpy(frozen["tem_obj_pts"]), torch.from_numpy(frozen["tem_v1"])
        assert ((pts_cam - t0[:, None]) @ R0 - obj_pts).abs().max() < 1e-5      # the stored points ARE this construction
    else:
        t0 = pts_cam.mean(dim=1)
        R0 = torch.linalg.qr(torch.randn(len(base), 3, 3, generator=gen))[0]
        R0 = R0 * torch.sign(torch.linalg.det(R0)).view(-1, 1, 1)               # proper rotations
        obj_pts = (pts_cam - t0[:, None]) @ R0                                 # p_cam = p_obj R0^T + t0
        v1 = obj_pts[:, 600:] + 2e-4 * torch.randn(len(base), 1448, 3, generator=gen)
    tem_pts = [obj_pts[:, :1400].contiguous(), v1.contiguous()]
    tem_choose = [choose[:, :1400].contiguous(), choose[:, 600:].contiguous()]
    models = obj_pts[:, :1024].contiguous()
    radius = models.norm(dim=2).max(dim=1).values
    keys = torch.rand(c["n_keys"], H * W, generator=torch.Generator().manual_seed(c["key_seed"]))
    return dict(fi=fi, tem_rgb=[crop, crop.clone()], tem_pts=tem_pts, tem_choose=tem_choose, model=models, radius=radius, keys=keys,
                gt_R=R0, gt_t=t0, tem_obj_pts=obj_pts, tem_v1=v1)
