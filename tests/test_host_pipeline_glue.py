"""FramePipeline's glue on a host without a device: the five models are replaced by recording stand-ins, everything
between them (Pillow-exact resize, small-detection filter, crop validity filter, top-k by final score, PEM pre-processing,
template expansion, result records) is the real code."""
import numpy as np
import torch

from sam6d_amd import pipeline
from sam6d_amd.sam.transforms import ResizeLongestSide


def test_glue_between_the_stages(monkeypatch):
    H, W, S = 120, 160, 256
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
    depth = 0.8 + 0.05 * torch.rand(H, W, generator=g)
    K = torch.tensor([[143.0, 0, 80.0], [0, 143.0, 60.0], [0, 0, 1]], dtype=torch.float64)
    seen = {}

    def fake_preprocess(x, img_size):
        seen["pre_in"] = x.clone()
        return torch.zeros(x.shape[0], 3, img_size, img_size)
    monkeypatch.setattr(pipeline, "sam_preprocess", fake_preprocess)
    monkeypatch.setattr(pipeline.FramePipeline, "_tick", lambda self, name, t0: self.times.__setitem__(name, 0.0) or t0)

    class Enc:
        img_size = S

        def __call__(self, x):
            seen["enc_in"] = tuple(x.shape)
            return torch.zeros(x.shape[0], 8, S // 16, S // 16)
    # six proposals: a tiny one (box below 0.05^2 of the frame), an exactly-square crop the reference cannot process, and
    # four usable ones
    boxes = torch.tensor([[2, 2, 6, 6], [10, 10, 59, 59], [20, 10, 90, 70], [60, 30, 150, 110], [5, 50, 70, 115], [80, 5, 155, 60]])
    masks = torch.zeros(6, H, W, dtype=torch.bool)
    for i, (x1, y1, x2, y2) in enumerate(boxes.tolist()):
        masks[i, y1:y2 + 1, x1:x2 + 1] = True

    def fake_proposals(pe, md, emb, orig, img_size, points_per_batch=64, **kw):
        seen["prop_args"] = (tuple(emb.shape), orig, img_size, points_per_batch, kw)
        return dict(masks=masks, boxes=boxes)
    monkeypatch.setattr(pipeline.amg, "generate_proposals", fake_proposals)

    class Desc:
        proposal_size = 56

        def __call__(self, image_np, prop):
            seen["desc_n"] = prop.masks.shape[0]
            assert torch.is_tensor(image_np) and tuple(image_np.shape) == (H, W, 3) and image_np.dtype == torch.uint8   # the device frame itself
            n = prop.masks.shape[0]
            return torch.zeros(n, 4), torch.zeros(n, 16, 4)

    class Scorer:
        def score(self, cls, patch, m, b, d, k, depth_scale=1.0):
            seen["depth_scale"] = depth_scale
            n = m.shape[0]
            final = torch.linspace(0.2, 0.9, n)                                # best = last surviving proposal
            return dict(final=final, sel=torch.arange(n), pred_obj=torch.zeros(n, dtype=torch.long))

    class Pem:
        def __call__(self, ep):
            seen["pem_shapes"] = {k: tuple(v.shape) for k, v in ep.items()}
            M = ep["pts"].shape[0]
            return dict(pred_R=torch.eye(3).expand(M, 3, 3).clone(), pred_t=torch.zeros(M, 3), pred_pose_score=torch.full((M,), 0.5))
    tpl = dict(model=torch.zeros(1, 64, 3), dense_po=torch.zeros(1, 32, 3), dense_fo=torch.zeros(1, 32, 8))
    pipe = pipeline.FramePipeline(Enc(), None, None, Desc(), Scorer(), Pem(), tpl, object_radius=10.0, top_k=3, points_per_batch=16,
                                  segmentor=dict(points_per_side=4))
    keys = torch.rand(3, H * W, generator=g)
    det, poses = pipe(img, depth, K, keys, torch.rand(3, 18000, generator=g))
    # resize: the encoder saw Pillow's pixels of the frame, as float, channels first
    ref = ResizeLongestSide(S).apply_image(img.numpy())
    assert seen["pre_in"].shape == (1, 3) + ref.shape[:2] and seen["pre_in"].dtype == torch.float32
    assert np.array_equal(seen["pre_in"][0].permute(1, 2, 0).numpy(), ref.astype(np.float32))
    assert seen["prop_args"][1:4] == ((H, W), S, 16) and seen["prop_args"][4] == dict(points_per_side=4)
    # filters: the 5 x 5 box is below 0.05^2 of the frame, the 50 x 50 crop is one the reference's CropResizePad rejects
    assert seen["desc_n"] == 4
    # top-k by final score, best first
    assert det.masks.shape == (3, H, W) and torch.equal(det.boxes, boxes[[5, 4, 3]])
    assert torch.equal(det.scores, torch.linspace(0.2, 0.9, 4).flip(0)[:3])
    M = poses["pred_R"].shape[0]
    assert M == 3 and poses["kept"].tolist() == [0, 1, 2]
    sh = seen["pem_shapes"]
    assert sh["pts"] == (3, 2048, 3) and sh["rgb"] == (3, 3, 224, 224) and sh["rgb_choose"] == (3, 2048)
    assert sh["model"] == (3, 64, 3) and sh["dense_po"] == (3, 32, 3) and sh["dense_fo"] == (3, 32, 8) and sh["coarse_rand_u"] == (3, 18000)
    out = pipeline.frame_results(det, poses, "ycbv", 0.1)
    assert len(out["ism_records"]) == 3 and len(out["csv_lines"]) == 3 and out["pem_records"][0]["R"] == np.eye(3).tolist()
    assert [r["category_id"] for r in out["ism_records"]] == [1, 1, 1]
    # ---- BOP-flow options: NMS per object id after scoring, detection score threshold, no top-k ---------------------------
    from oracle import sam_decoder as osd
    from sam6d_amd.ism import handoff
    monkeypatch.setattr(handoff, "_device_nms", osd.nms)
    boxes[4] = torch.tensor([62, 32, 150, 110])                      # now overlaps proposal 3 (IoU > 0.25): the weaker one goes
    masks[4] = False
    masks[4, 32:111, 62:151] = True
    pipe = pipeline.FramePipeline(Enc(), None, None, Desc(), Scorer(), Pem(), tpl, object_radius=10.0, top_k=None, points_per_batch=16,
                                  nms_per_object_thresh=0.25, det_score_thresh=0.3)
    det, poses = pipe(img, depth, K, torch.rand(4, H * W, generator=g), torch.rand(4, 18000, generator=g))
    # survivors of the filters in proposal order: 2, 3, 4, 5 with final scores .2, .433, .667, .9; NMS drops 3 (overlaps 4,
    # lower score); the score threshold drops 2; best first
    assert torch.equal(det.boxes, boxes[[5, 4]]) and torch.allclose(det.scores, torch.tensor([0.9, 0.2 + 0.7 * 2 / 3]))
    assert poses["pred_R"].shape[0] == 2
    pipe.det_thresh = 0.95
    det, poses = pipe(img, depth, K, torch.rand(4, H * W, generator=g), torch.rand(4, 18000, generator=g))
    assert len(det) == 0 and poses is None
    assert seen["depth_scale"] == 1000.0                  # metres in, the ISM's millimetre contract out
    # ---- several objects: template rows and radius follow the predicted object id ------------------------------------------
    class Scorer2(Scorer):
        def score(self, cls, patch, m, b, d, k, depth_scale=1.0):
            r = super().score(cls, patch, m, b, d, k, depth_scale)
            r["pred_obj"] = torch.tensor([2, 0, 1, 2])[: m.shape[0]]
            return r
    tpl3 = dict(model=torch.arange(3.0)[:, None, None].expand(3, 64, 3).clone(), dense_po=torch.arange(3.0)[:, None, None].expand(3, 32, 3).clone(),
                dense_fo=torch.zeros(3, 32, 8))

    class Pem2(Pem):
        def __call__(self, ep):
            seen["model_rows"] = ep["model"][:, 0, 0].tolist()
            seen["po_rows"] = ep["dense_po"][:, 0, 0].tolist()
            return super().__call__(ep)
    pipe = pipeline.FramePipeline(Enc(), None, None, Desc(), Scorer2(), Pem2(), tpl3, object_radius=torch.tensor([10.0, 10.0, 10.0]),
                                  top_k=None, points_per_batch=16)
    det, poses = pipe(img, depth, K, torch.rand(4, H * W, generator=g), torch.rand(4, 18000, generator=g))
    assert det.object_ids.tolist() == [2, 1, 0, 2]                   # best first: proposals 5, 4, 3, 2 -> objects 2, 1, 0, 2
    assert seen["model_rows"] == [2.0, 1.0, 0.0, 2.0] == seen["po_rows"]
    out = pipeline.frame_results(det, poses, "ycbv", 0.0)
    assert [r["category_id"] for r in out["pem_records"]] == [3, 2, 1, 3] and [l.split(",")[2] for l in out["csv_lines"]] == ["3", "2", "1", "3"]
    # ---- a group of frames: ONE encoder pass and ONE PEM pass; the middle frame has no valid depth (nothing survives its
    # pre-processing), so the PEM batch is frames 0 and 2 back to back and is split again per frame
    def rnd():
        return torch.rand(4, H * W, generator=g), torch.rand(4, 18000, generator=g)
    (k0, u0), (k1, u1), (k2, u2) = rnd(), rnd(), rnd()
    res = pipe.run_group([(img, depth, K, k0, u0), (img, torch.zeros_like(depth), K, k1, u1), (img, depth, K, k2, u2)])
    assert seen["enc_in"][0] == 3 and len(res) == 3
    assert seen["model_rows"] == [2.0, 1.0, 0.0, 2.0] * 2 and seen["pem_shapes"]["pts"] == (8, 2048, 3)
    assert seen["pem_shapes"]["coarse_rand_u"] == (8, 18000)
    (d0, p0), (d1, p1), (d2, p2) = res
    assert len(d1) == 4 and p1 is None
    assert p0["pred_R"].shape == (4, 3, 3) and p2["pred_t"].shape == (4, 3) and p0["kept"].tolist() == [0, 1, 2, 3] == p2["kept"].tolist()
    assert torch.equal(d0.boxes, d2.boxes) and d0.object_ids.tolist() == [2, 1, 0, 2]
    alone = pipe(img, depth, K, k2, u2)
    assert torch.equal(alone[0].boxes, d2.boxes) and torch.equal(alone[1]["pred_R"], p2["pred_R"])



def _known_answer_scene():
    """One object (a 4 cm ball of model points, identity template pose) seen at 0.8 m in the middle of a box mask."""
    H, W = 120, 160
    g = torch.Generator().manual_seed(3)
    K = torch.tensor([[143.0, 0, 80.0], [0, 143.0, 60.0], [0, 0, 1]], dtype=torch.float64)
    pc = torch.nn.functional.normalize(torch.randn(1, 512, 3, generator=g), dim=-1) * 0.04
    poses = torch.eye(4)[None]
    C = 32
    ref_cls = torch.randn(1, 1, C, generator=g)
    ref_patch = torch.nn.functional.normalize(torch.randn(1, 1, 16, C, generator=g), dim=-1)
    masks = torch.zeros(1, H, W)
    masks[0, 50:71, 70:91] = 1.0                                    # 21 x 21 px around the principal point
    boxes = torch.tensor([[70.0, 50.0, 90.0, 70.0]])
    depth_m = torch.full((H, W), 0.8)
    return H, W, K, pc, poses, ref_cls, ref_patch, masks, boxes, depth_m


def test_depth_unit_boundary_known_answer():
    """the pipeline takes depth in metres, the ISM translation works on millimetres x depth_scale / 1000.
    A known-answer object: the query translation must be the mask's depth (0.8 m, not 0.0008) and the projected template
    box must overlap the proposal box."""
    from sam6d_amd.ism.scoring import FrameScorer
    H, W, K, pc, poses, ref_cls, ref_patch, masks, boxes, depth_m = _known_answer_scene()
    scorer = FrameScorer(ref_cls, ref_patch, poses, pc, confidence_thresh=-1.0, aggregation_function="max")
    pipe = pipeline.FramePipeline.__new__(pipeline.FramePipeline)
    pipe.scorer = scorer
    sc = pipe.score_metres(ref_cls[0], ref_patch[0], masks, boxes, depth_m, K)
    t = scorer.Calculate_the_query_translation(masks.clone(), depth_m, K, 1000.0)
    assert abs(t[0, 2].item() - 0.8) < 1e-6 and abs(t[0, 0].item()) < 5e-3 and abs(t[0, 1].item()) < 5e-3
    uv = sc["image_uv"][0]
    # a 4 cm ball at 0.8 m spans 143 * 0.04 / 0.8 = 7 px either side of the principal point
    assert 70 <= uv[:, 0].min() <= 76 and 84 <= uv[:, 0].max() <= 90 and 50 <= uv[:, 1].min() <= 56
    assert torch.is_tensor(sc["iou"]) and sc["iou"][0] > 0.3
    # the same frame in the reference's own unit gives the same answer
    sc_mm = scorer.score(ref_cls[0], ref_patch[0], masks, boxes, depth_m * 1000.0, K, depth_scale=1.0)
    assert torch.equal(sc_mm["image_uv"], sc["image_uv"])
    # and the old call (metres with depth_scale 1) is the failure the advisor described: every point lands on one pixel
    bad = scorer.score(ref_cls[0], ref_patch[0], masks, boxes, depth_m, K, depth_scale=1.0)
    assert not torch.is_tensor(bad["iou"]) or bad["iou"][0] < 0.05


def test_pipeline_rejects_too_few_random_rows():
    import pytest
    pipe = pipeline.FramePipeline(None, None, None, None, None, None, {}, 1.0, top_k=4)
    with pytest.raises(ValueError, match="one row per detection"):
        pipe(torch.zeros(8, 8, 3, dtype=torch.uint8), torch.zeros(8, 8), torch.eye(3), torch.zeros(2, 64), torch.zeros(4, 18000))


def test_descriptor_batches_fill_the_tile_grid():
    """dinov2.plan_chunks: the batch sizes add up, no batch exceeds 255 proposals (256 row tiles), full groups are cut into 255s
    (255 x 257 token rows = 256 row tiles exactly) and the plan never costs more tile rounds than one batch per 128 proposals."""
    from sam6d_amd.ism.dinov2 import plan_chunks

    def rounds(c):                                                      # proj / fc2: 4 column tiles per row tile on 256 CUs
        mt = -(-c * 257 // 256)
        return -(-mt * 4 // 256)
    for n in (1, 7, 63, 64, 128, 255, 256, 300, 640, 1024, 1531):
        plan = plan_chunks(n)
        assert sum(plan) == n and max(plan) <= 255 and min(plan) >= 1
        naive = [128] * (n // 128) + ([n % 128] if n % 128 else [])
        assert sum(rounds(c) for c in plan) <= sum(rounds(c) for c in naive)
    assert plan_chunks(1024) == [255, 255, 255, 255, 4]
    assert plan_chunks(128) == [128]


def test_layernorm_fold_algebra_on_the_host():
    """utils.linear.lnfold_weights: LN(x) W^T + b == ((x W'^T) + sigma b' - mean s) / sigma with W' = gamma * W (rounded to bf16: the
    identity is checked on the ROUNDED weight, which is what the kernel multiplies), s = row sums of W', b' = b + W beta -- the
    statement s6d_gemm_bf16_lnfold implements, here in float64 on the CPU."""
    import torch

    from sam6d_amd.utils.linear import lnfold_weights
    g = torch.Generator().manual_seed(0)
    M, K, N = 37, 96, 24
    x = torch.randn(M, K, generator=g, dtype=torch.float64) * 3 + 5
    W = torch.randn(N, K, generator=g, dtype=torch.float64) / K ** 0.5
    b = torch.randn(N, generator=g, dtype=torch.float64)
    gamma = 1 + 0.3 * torch.randn(K, generator=g, dtype=torch.float64)
    beta = 0.2 * torch.randn(K, generator=g, dtype=torch.float64)
    wf, cs, bf = lnfold_weights(W.float(), b.float(), gamma.float(), beta.float())
    mean = x.mean(1, keepdim=True)
    sigma = torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-6)
    folded = (x @ wf.double().t() + sigma * bf.double() - mean * cs.double()) / sigma
    # the same LayerNorm -> Linear with the rounded folded weight spelled out: ((x - mean) / sigma) @ W'^T + b'
    direct = ((x - mean) / sigma) @ wf.double().t() + bf.double()
    assert (folded - direct).abs().max() < 1e-9
    # and against the unrounded statement: only the bf16 rounding of gamma * W separates them
    true = torch.nn.functional.layer_norm(x, (K,), gamma, beta, 1e-6) @ W.t() + b
    assert ((direct - true).pow(2).mean() / true.pow(2).mean()).sqrt() < 3e-3
    assert wf.dtype == torch.bfloat16 and cs.dtype == torch.float32 and bf.dtype == torch.float32
