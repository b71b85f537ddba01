"""Where the PEM per-detection pre-processing (sam6d_amd/pem/preprocess.py) spends its time: the tensor ops of its library-op
path grouped into stages with a device synchronisation after each, then whole calls of the kernel path (the default) and of the
library-op path.  Round 1: 107 ms for 64 detections, 104 of them a float64 index_add_ centroid (deleted); round 2: 3.9 ms kernel
path, 6.0 ms library-op path (profiles/r02_pem_pre_time.txt)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sam6d_amd.pem import preprocess as pre  # noqa: E402
from sam6d_amd.utils import synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inp = synth.pem_pre_inputs(P=P, seed=9)
dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")   # cpu: dry run of the script only
image, depth, K, masks, keys = (torch.from_numpy(inp["image"]).to(dev), inp["depth"].to(dev), inp["K"], inp["masks"].to(dev),
                                inp["keys"].to(dev))
radius, n_sample, img_size = 0.15, 2048, 224
marks = []


def tick(name):
    if dev.type == "cuda":
        torch.cuda.synchronize()
    marks.append((name, time.perf_counter()))


for it in range(3):
    marks.clear()
    tick("start")
    Pn, H, W = masks.shape
    m = (masks > 0) & (depth > 0)[None]
    cnt = m.flatten(1).sum(1)
    ok1 = cnt > 32
    box = pre.square_boxes(m | ~ok1[:, None, None])
    y1, y2, x1, x2 = box.unbind(1)
    tick("mask AND depth, counts, square boxes")
    pyx = torch.nonzero(m)
    tick("nonzero (host round trip)")
    p_, y_, x_ = pyx.unbind(1)
    inside = (y_ >= y1[p_]) & (y_ < y2[p_]) & (x_ >= x1[p_]) & (x_ < x2[p_]) & ok1[p_]
    p_, y_, x_ = p_[inside], y_[inside], x_[inside]
    choose = (y_ - y1[p_]) * (x2 - x1)[p_] + (x_ - x1[p_])
    z = depth[y_, x_]
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    fx_t, fy_t = torch.tensor(fx, device=dev), torch.tensor(fy, device=dev)
    cloud = torch.stack([(x_.float() - cx) * z / fx_t, (y_.float() - cy) * z / fy_t, z], 1)
    tick("inside-box filter, crop indices, back-projection")
    n0 = torch.bincount(p_, minlength=Pn)
    center = pre._segment_seq_sum(cloud.contiguous(), (torch.cumsum(n0, 0) - n0).contiguous(), n0.contiguous()) / n0.clamp(min=1).float()[:, None]
    dist = torch.linalg.norm(cloud - center[p_], dim=1)
    flag = dist.double() < radius * 1.2
    p_, choose, cloud = p_[flag], choose[flag], cloud[flag]
    n = torch.bincount(p_, minlength=Pn)
    tick("centroid, radius filter")
    idx = pre._keyed_indices(n, keys, n_sample)
    tick("sampler (composite-key top-k)")
    ok = ok1 & (n >= 4)
    start = torch.cumsum(n, 0) - n
    kept = torch.nonzero(ok).squeeze(1)
    g = (start[:, None] + idx)[kept].clamp(max=max(cloud.shape[0] - 1, 0))
    pts, ch = cloud[g], choose[g]
    tick("gather sampled points")
    bk = box[kept]
    rgb = pre._crops(image, m[kept].float(), bk, img_size, True)
    tick("colour crops (bilinear, normalise)")
    if it == 2:
        t0 = marks[0][1]
        print(f"P = {P}, kept {len(kept)}, list length {pyx.shape[0]}")
        for (a, ta), (_, tb) in zip(marks[1:], marks[:-1]):
            print(f"  {a:52s} {(ta - tb) * 1e3:8.2f} ms")
        print(f"  {'total':52s} {(marks[-1][1] - t0) * 1e3:8.2f} ms")

# ---- whole calls: the kernel path (default on the device) and the library-op path -------------------------------------------------
if dev.type == "cuda":
    import os
    for tag, env in (("kernel path (default)", None), ("library-op path (S6D_PEM_PRE=library)", "library")):
        if env is None:
            os.environ.pop("S6D_PEM_PRE", None); __import__("sam6d_amd.policy").policy.reload()
        else:
            os.environ["S6D_PEM_PRE"] = env; __import__("sam6d_amd.policy").policy.reload()
        pre.observed_inputs(image, depth, K, masks, radius, keys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            pre.observed_inputs(image, depth, K, masks, radius, keys)
        torch.cuda.synchronize()
        print(f"observed_inputs, {tag}, P = {P}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
    os.environ.pop("S6D_PEM_PRE", None); __import__("sam6d_amd.policy").policy.reload()
