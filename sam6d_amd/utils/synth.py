"""Seeded synthetic inputs for the SAM-6D per-frame hot path.

No BOP data, templates or checkpoints are reachable offline, so parity tests,
golden generation and bench.py all draw their inputs here.  Everything is
generated on the CPU with an explicit torch.Generator, so the same arguments
give the same tensors in the build container and on the GPU box.
"""

import torch


def _g(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def random_rotations(B, g):
    """Haar-distributed rotations from normalised quaternions."""
    q = torch.randn(B, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)
    return R.reshape(B, 3, 3)


def pem_inputs(B, seed=1, n_pts=2048, n_model=1024, feat_dim=256, img=224, with_rgb=True):
    """One PEM batch (Net.forward keys: pts, rgb, rgb_choose, model, dense_po, dense_fo)
    plus the ground truth of the synthetic rigid motion and a known-answer feature set.

    dense_po: n_pts points uniform in a ball of radius 0.1 m (object frame);
    pts = dense_po R^T + t + N(0, 1e-3 * 0.1);  t = (0.1, -0.05, 0.8);
    dense_fo ~ N(0,1); ``dense_fm_kat`` = dense_fo + 0.1 N(0,1) is the known-answer
    observed feature (bypasses the ViT): with it the matcher must recover (R, t).
    """
    g = _g(seed)
    d = torch.randn(B, n_pts, 3, generator=g)
    d = d / d.norm(dim=2, keepdim=True)
    r = torch.rand(B, n_pts, 1, generator=g) ** (1.0 / 3.0)
    dense_po = 0.1 * d * r
    R = random_rotations(B, g)
    t = torch.tensor([0.1, -0.05, 0.8]).expand(B, 3).contiguous()
    pts = dense_po @ R.transpose(1, 2) + t.unsqueeze(1) + 1e-4 * torch.randn(B, n_pts, 3, generator=g)
    dense_fo = torch.randn(B, n_pts, feat_dim, generator=g)
    dense_fm_kat = dense_fo + 0.1 * torch.randn(B, n_pts, feat_dim, generator=g)
    model = dense_po[:, torch.randperm(n_pts, generator=g)[:n_model]].contiguous()
    out = dict(pts=pts.contiguous(), model=model, dense_po=dense_po.contiguous(), dense_fo=dense_fo,
               gt_R=R, gt_t=t, dense_fm_kat=dense_fm_kat)
    if with_rgb:
        out["rgb"] = torch.randn(B, 3, img, img, generator=g)
        out["rgb_choose"] = torch.randint(0, img * img, (B, n_pts), generator=g)
    return out


def coarse_uniforms(B, seed=1, n=18000):
    """The uniform samples compute_coarse_Rt draws (model_utils.py:219): an INPUT here.
    Equal to ``torch.manual_seed(seed); torch.rand(B, n)`` on the CPU generator."""
    return torch.rand(B, n, generator=_g(seed))


def sam_input(B=1, seed=1, size=1024):
    return torch.randn(B, 3, size, size, generator=_g(seed))


def ism_inputs(P=128, O=8, T=42, C=1024, n_patch=256, H=480, W=640, seed=1, n_model_pts=2048):
    """Proposal / reference descriptors and masks for the ISM scoring path (a6-a9)."""
    g = _g(seed)
    ref_cls = torch.randn(O, T, C, generator=g)
    # make every proposal resemble one (object, template) so that thresholds are exercised
    obj = torch.randint(0, O, (P,), generator=g)
    tem = torch.randint(0, T, (P,), generator=g)
    mix = torch.rand(P, 1, generator=g)
    qry_cls = mix * ref_cls[obj, tem] + (1 - mix) * torch.randn(P, C, generator=g)
    ref_patch = torch.nn.functional.normalize(torch.randn(O, T, n_patch, C, generator=g), dim=-1)
    ref_patch = ref_patch * (torch.rand(O, T, n_patch, 1, generator=g) > 0.3)
    qry_patch = torch.nn.functional.normalize(
        0.7 * ref_patch[obj, tem] + 0.3 * torch.nn.functional.normalize(torch.randn(P, n_patch, C, generator=g), dim=-1),
        dim=-1)
    qry_patch = qry_patch * (torch.rand(P, n_patch, 1, generator=g) > 0.3)
    # axis-aligned box masks and boxes (xyxy)
    x0 = torch.randint(0, W - 80, (P,), generator=g)
    y0 = torch.randint(0, H - 80, (P,), generator=g)
    bw = torch.randint(24, 80, (P,), generator=g)
    bh = torch.randint(24, 80, (P,), generator=g)
    ys = torch.arange(H).view(1, H, 1)
    xs = torch.arange(W).view(1, 1, W)
    masks = ((ys >= y0.view(-1, 1, 1)) & (ys < (y0 + bh).view(-1, 1, 1)) &
             (xs >= x0.view(-1, 1, 1)) & (xs < (x0 + bw).view(-1, 1, 1))).float()
    boxes = torch.stack([x0, y0, x0 + bw - 1, y0 + bh - 1], 1).float()
    depth = (1000.0 + 50.0 * torch.rand(H, W, generator=g)).round()
    depth[torch.rand(H, W, generator=g) < 0.05] = 0  # holes
    K = torch.tensor([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=torch.float64)
    pc = 0.05 * torch.randn(O, n_model_pts, 3, generator=g)
    poses = torch.eye(4).repeat(T, 1, 1)
    poses[:, :3, :3] = random_rotations(T, g)
    return dict(qry_cls=qry_cls, ref_cls=ref_cls, qry_patch=qry_patch, ref_patch=ref_patch, masks=masks,
                boxes=boxes, depth=depth, K=K, pointcloud=pc, poses=poses, gt_obj=obj, gt_tem=tem)


def dinov2_inputs(P=8, H=480, W=640, seed=1):
    """RGB frame (H,W,3) uint8 + proposal masks (P,H,W) float {0,1} + xyxy long boxes for the DINOv2 crop path.
    Proposals 0..4 pin the branches of CropResizePad: square 100x100 crop (no padding; most other square sizes make
    the reference itself raise in torch.stack, its second resize flooring to target-1), very tall, very wide, tiny, near-full
    frame; the rest are random.  Masks are ellipses inscribed in their boxes (so masked-out pixels occur inside
    the crop), boxes are the mask extents as SAM's batched_mask_to_box reports them (inclusive max corner)."""
    g = _g(seed)
    img = (torch.rand(H // 8, W // 8, 3, generator=g) * 255).repeat_interleave(8, 0).repeat_interleave(8, 1)
    img = (img + 20 * torch.randn(H, W, 3, generator=g)).clamp(0, 255).to(torch.uint8)
    fixed = [(40, 30, 101, 101), (200, 10, 24, 300), (10, 400, 500, 30), (320, 240, 5, 4), (2, 3, W - 6, H - 8)]
    rects = []
    for i in range(P):
        if i < len(fixed):
            rects.append(fixed[i])
        else:
            bw = int(torch.randint(12, W // 2, (1,), generator=g))
            bh = int(torch.randint(12, H // 2, (1,), generator=g))
            rects.append((int(torch.randint(0, W - bw, (1,), generator=g)), int(torch.randint(0, H - bh, (1,), generator=g)), bw, bh))
    ys = torch.arange(H).view(H, 1).float()
    xs = torch.arange(W).view(1, W).float()
    masks, boxes = [], []
    for x0, y0, bw, bh in rects:
        cx, cy = x0 + (bw - 1) / 2, y0 + (bh - 1) / 2
        m = (((xs - cx) / (bw / 2)) ** 2 + ((ys - cy) / (bh / 2)) ** 2 <= 1.0).float()
        yy, xx = torch.nonzero(m, as_tuple=True)
        masks.append(m)
        boxes.append(torch.stack([xx.min(), yy.min(), xx.max(), yy.max()]))
    return dict(image=img.numpy(), masks=torch.stack(masks), boxes=torch.stack(boxes).long())


def random_boxes(P, H, W, g):
    """xyxy long boxes of every aspect ratio with sides from 2 px to the full frame, minus the shapes on which the
    reference's CropResizePad itself raises (square crops, sides that vanish after the resize)."""
    x1 = torch.randint(0, W - 3, (P,), generator=g)
    y1 = torch.randint(0, H - 3, (P,), generator=g)
    x2 = (x1 + 2 + (torch.rand(P, generator=g) ** 2 * (W - x1 - 2)).long()).clamp(max=W)
    y2 = (y1 + 2 + (torch.rand(P, generator=g) ** 2 * (H - y1 - 2)).long()).clamp(max=H)
    boxes = torch.stack([x1, y1, x2, y2], 1)
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    ok = (w != h) & (torch.minimum(w, h).float() * 224 / torch.maximum(w, h).float() >= 1.5)
    return boxes[ok]


def sam_decoder_inputs(cfg, n_prompts, seed=1):
    """Image embedding (1,C,h,w) like the encoder neck's output (LayerNorm2d-normalised scale), point prompts in
    input-image pixels (a regular grid like the automatic mask generator's, plus label-0 and two-point cases) and
    xyxy box prompts."""
    g = _g(seed)
    emb = torch.randn(1, cfg["dim"], cfg["emb"], cfg["emb"], generator=g)
    side = int(n_prompts ** 0.5) or 1
    off = 1.0 / (2 * side)
    gx = torch.linspace(off, 1 - off, side)
    grid = torch.stack(torch.meshgrid(gx, gx, indexing="xy"), -1).reshape(-1, 2)[:n_prompts] * cfg["img"]
    if grid.shape[0] < n_prompts:
        grid = torch.cat([grid, torch.rand(n_prompts - grid.shape[0], 2, generator=g) * cfg["img"]])
    points = grid.unsqueeze(1)                                       # (B,1,2): one foreground point per prompt
    labels = torch.ones(n_prompts, 1)
    points2 = torch.rand(3, 2, 2, generator=g) * cfg["img"]          # two points, second one background
    labels2 = torch.tensor([[1.0, 0.0]]).repeat(3, 1)
    xy = torch.rand(3, 2, generator=g) * cfg["img"] * 0.5
    boxes = torch.cat([xy, xy + 0.1 * cfg["img"] + torch.rand(3, 2, generator=g) * cfg["img"] * 0.4], dim=1)
    return dict(emb=emb, points=points, labels=labels, points2=points2, labels2=labels2, boxes=boxes)


def sam_lowres_logits(B, C=3, n=256, seed=1):
    """Low-resolution mask logits (B,C,n,n) with the look of the decoder's output: smooth blobs of logit ~ +-8 with
    soft edges (so that the +-1 stability band and the zero crossing both cut through many pixels), one empty mask
    (all negative) and one full mask."""
    g = _g(seed)
    ys, xs = torch.meshgrid(torch.arange(n).float(), torch.arange(n).float(), indexing="ij")
    out = torch.empty(B, C, n, n)
    for b in range(B):
        for c in range(C):
            cx, cy = torch.rand(2, generator=g) * n
            sx, sy = 8 + torch.rand(2, generator=g) * n / 4
            blob = 14 * torch.exp(-((xs - cx) ** 2 / (2 * sx ** 2) + (ys - cy) ** 2 / (2 * sy ** 2))) - 6
            out[b, c] = blob + 0.3 * torch.randn(n, n, generator=g)
    out[0, 0] = -5 + 0.3 * torch.randn(n, n, generator=g)
    out[-1, -1] = 5 + 0.3 * torch.randn(n, n, generator=g)
    return out


def pem_pre_inputs(P=8, H=480, W=640, seed=1):
    """A frame for the PEM pre-processing: RGB image, proposal masks (dinov2_inputs' shapes: square, tall, wide, tiny,
    near-full, random), a depth map in metres with holes whose surface bulges under every mask (so the radius filter
    cuts some points), the LM camera, and one uniform key per pixel and detection for the sampler."""
    d = dinov2_inputs(P=P, H=H, W=W, seed=seed)
    g = _g(seed + 100)
    ys = torch.arange(H).view(H, 1).float()
    xs = torch.arange(W).view(1, W).float()
    depth = 0.9 + 0.1 * torch.sin(xs / 37.0) * torch.cos(ys / 29.0) + 0.002 * torch.randn(H, W, generator=g)
    depth[torch.rand(H, W, generator=g) < 0.05] = 0.0
    K = torch.tensor([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=torch.float64)
    keys = torch.rand(P, H * W, generator=g)
    return dict(image=d["image"], masks=d["masks"] > 0, depth=depth, K=K, keys=keys)
