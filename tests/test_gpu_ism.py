"""GPU parity of the ISM scoring chain vs the reference golden."""
import ast

import numpy as np
import pytest
import torch

from sam6d_amd.utils import synth
from tests import util

pytestmark = pytest.mark.gpu


def _inputs(g):
    c = ast.literal_eval(str(g["case"]))
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in
            synth.ism_inputs(P=c["P"], O=c["O"], T=c["T"], seed=c["seed"]).items()}


# P=64 / O=3, the benched shape P=128 / O=1 (BASELINE configs[1]), YCB-V P=128 / O=21 (configs[2]), T-LESS P=256 / O=30 (configs[3])
GOLDENS = ["ism_scoring.npz", "ism_scoring_p128.npz", "ism_scoring_ycbv.npz", "ism_scoring_tless.npz"]


def _check_pairwise(got, g):
    """Small fixtures hold the (P, O*T) cosine matrix whole, the large ones a strided sample + sums (gen_golden.digest)."""
    if "pairwise" in g.files:
        np.testing.assert_allclose(got.cpu().numpy(), g["pairwise"], atol=2e-6)
    else:
        flat = got.detach().double().reshape(-1).cpu()
        np.testing.assert_allclose(flat[::13].float().numpy(), g["pairwise_sample"], rtol=0, atol=2e-6)
        assert abs(flat.abs().sum().item() - g["pairwise_sums"][1]) <= 2e-6 * flat.numel()


@pytest.mark.parametrize("name", GOLDENS)
def test_projection_is_bit_exact_given_the_reference_translation(name):
    """a9, integer part: project_template_to_image (detector.py:209-232) fed the reference's own query translation gives
    the reference's pixels and boxes BIT FOR BIT (the kernel spells out the op order of the pinned run: k-ascending fma
    chains, rounded add, IEEE division, truncation), and with equal boxes the IoU agrees to float rounding."""
    from sam6d_amd import ops
    from sam6d_amd.ism.scoring import compute_iou
    g = util.golden(name)
    inp = _inputs(g)
    sel = torch.from_numpy(g["sel"]).cuda()
    uv, bbox = ops.project_bbox(inp["pointcloud"].contiguous(), inp["poses"].contiguous(),
                                torch.from_numpy(g["pred_obj"]).int().cuda(), torch.from_numpy(g["best_template"]).int().cuda(),
                                torch.from_numpy(np.ascontiguousarray(g["translation"])).cuda(), inp["K"].to(torch.float32).cuda().contiguous(),
                                inp["depth"].shape[0], inp["depth"].shape[1])
    assert np.array_equal(uv.cpu().numpy(), g["image_uv"])
    ref_box = np.concatenate((g["image_uv"].min(1), g["image_uv"].max(1)), -1)
    assert np.array_equal(bbox.cpu().numpy(), ref_box)
    iou = compute_iou(bbox, inp["boxes"][sel])
    np.testing.assert_allclose(torch.as_tensor(iou).cpu().numpy(), g["iou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(compute_iou(bbox.float(), torch.from_numpy(g["boxes2"]).cuda()).cpu().numpy(), g["iou2"],
                               rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", GOLDENS)
def test_frame_scoring_vs_reference_golden(name):
    """The whole matching stage (a6-a9).  Indices exact; scores to float rounding; the query translation, the projected
    pixels and the boxes BIT FOR BIT.  The translation is a mean over ~10^4 pixels whose float32 / float64 sums carry their
    order: the kernel follows the reduction tree of the pinned reference run (ATen's CPU cascade sum, oracle/aten_sum.py),
    so the truncated pixel coordinates downstream are the reference's at every truncation border."""
    from sam6d_amd.ism.scoring import FrameScorer
    g = util.golden(name)
    inp = _inputs(g)
    fs = FrameScorer(inp["ref_cls"], inp["ref_patch"], inp["poses"], inp["pointcloud"])
    _check_pairwise(fs.matching_config.metric(inp["qry_cls"], inp["ref_cls"]), g)
    out = fs.score(inp["qry_cls"], inp["qry_patch"], inp["masks"], inp["boxes"], inp["depth"], inp["K"])
    for k in ("sel", "pred_obj", "best_template"):
        assert np.array_equal(out[k].cpu().numpy(), g[k]), k
    t = fs.Calculate_the_query_translation(inp["masks"][out["sel"]].clone(), inp["depth"], inp["K"], 1.0).cpu().numpy()
    assert t.dtype == np.float32 and np.array_equal(t, g["translation"])
    assert np.array_equal(out["image_uv"].cpu().numpy(), g["image_uv"])
    for k in ("semantic", "appearance", "visible_ratio"):
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=0, atol=1e-5, err_msg=k)
    box = torch.cat((out["image_uv"].min(1).values, out["image_uv"].max(1).values), -1).cpu().numpy()
    assert np.array_equal(box, np.concatenate((g["image_uv"].min(1), g["image_uv"].max(1)), -1))
    iou = torch.as_tensor(out["iou"]).cpu().numpy() * np.ones(len(box), np.float32)
    np.testing.assert_allclose(iou, g["iou"] * np.ones(len(box), np.float32), rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["final"].cpu().numpy(), g["final"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("H,W", [(480, 640), (120, 160), (37, 44), (130, 100), (2, 4), (96, 1028)])
def test_translation_sum_order_vs_oracle(H, W):
    """masked_depth_* against the numpy restatement of ATen's CPU sum order (oracle/aten_sum.py) BIT FOR BIT, at map sizes
    that exercise every remainder of the tree: partial level-1 / level-2 nodes, row remainder, tail vectors, scalar tail
    (37*44 = 1628 = 50 float rows + 3 vectors + 4 scalars), a map smaller than one level-0 block, frames in one launch."""
    from oracle import ism as oism
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(H * 1000 + W)
    S, F = 5, 2
    masks = (torch.rand(S, H, W, generator=g) > 0.6).float()
    masks[0] = 1.0                                            # a full-frame mask: every node of the tree is populated
    depth = torch.rand(F, H, W, generator=g) * 900 + 300
    depth[:, ::7, ::5] = 0                                    # invalid pixels inside the masks
    K = torch.tensor([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], dtype=torch.float64)
    Ks = torch.stack((K, K * torch.tensor([[1.1], [0.9], [1.0]], dtype=torch.float64)))
    frame = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32)
    out = ops.masked_depth_mean(masks.cuda(), depth.cuda(), Ks.cuda(), 1.0, frame=frame.cuda()).cpu().numpy()
    for s in range(S):
        want = oism.mean_translation_pinned(masks[s:s + 1], depth[frame[s]], Ks[frame[s]], 1.0).numpy()
        assert np.array_equal(out[s:s + 1], want), (s, out[s], want)
    one = ops.masked_depth_mean(masks.cuda(), depth[0].cuda(), K.cuda(), 1.0).cpu().numpy()
    assert np.array_equal(one, oism.mean_translation_pinned(masks, depth[0], K, 1.0).numpy())


def test_ism_kernels_individually_vs_oracle():
    """Each fused ISM kernel against the CPU oracle on odd, non-tile-aligned sizes."""
    from oracle import ism as oism
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(3)
    # cosine: P, R not multiples of 16
    q, r = torch.randn(37, 1024, generator=g), torch.randn(3, 45, 1024, generator=g)
    ref = oism.pairwise_similarity(q, r)
    out = ops.pairwise_cosine(q.cuda(), r.reshape(-1, 1024).cuda()).cpu().view(37, 3, 45)
    assert (out - ref).abs().max() < 2e-6
    # semantic select vs torch (avg_5 / max / mean)
    sc = torch.rand(37, 3, 45, generator=g)
    for topk in (5, 1, 45):
        bs, bo, bt = (t.cpu() for t in ops.semantic_select(sc.cuda(), topk))
        per = torch.topk(sc, k=topk, dim=-1)[0].mean(-1)
        es, eo = per.max(-1)
        assert torch.equal(bo.long(), eo) and (bs - es).abs().max() < 1e-6
        assert torch.equal(bt.long(), torch.gather(sc.argmax(-1), 1, eo[:, None])[:, 0])
    # patch scores: N1 = 200 rows (not a multiple of 64), N2 = 250 cols, zeroed rows/cols
    S, N1, N2, C = 5, 200, 250, 256
    store = torch.nn.functional.normalize(torch.randn(2, 3, N2, C, generator=g), dim=-1)
    store = store * (torch.rand(2, 3, N2, 1, generator=g) > 0.3)
    obj = torch.tensor([0, 1, 1, 0, 1])
    tm = torch.tensor([2, 0, 1, 1, 2])
    qp = torch.nn.functional.normalize(0.6 * store[obj, tm][:, :N1] + 0.4 * torch.randn(S, N1, C, generator=g), dim=-1)
    qp = qp * (torch.rand(S, N1, 1, generator=g) > 0.3)
    ea, eref = oism.appearance_score(qp, store, obj, tm)
    er = oism.visible_ratio(qp, eref, 0.5)
    a, rr = ops.patch_scores(qp.cuda(), store.cuda(), obj.int().cuda(), tm.int().cuda(), 0.5)
    assert (a.cpu() - ea).abs().max() < 2e-6 and (rr.cpu() - er).abs().max() < 2e-6


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_half_descriptors_take_the_kernels(dt, monkeypatch):
    """b3: the BOP flow runs under Lightning precision=16 (configs/machine/trainer/local.yaml:9): half descriptors must reach the
    same kernels (fp32 arithmetic on the rounded operands), not a library fallback.  Comparand: the oracle statements in fp32 on
    the SAME half-rounded descriptors."""
    from oracle import ism as oism
    from sam6d_amd import ops
    from sam6d_amd.ism.loss import MaskedPatch_MatrixSimilarity, PairwiseSimilarity
    d = synth.ism_inputs(P=12, O=3, T=6, C=128, n_patch=48, H=120, W=160, seed=5)
    calls = {"cos": 0, "patch": 0}
    oc, op = ops.pairwise_cosine, ops.patch_scores
    monkeypatch.setattr(ops, "pairwise_cosine", lambda *a, **k: (calls.__setitem__("cos", calls["cos"] + 1), oc(*a, **k))[1])
    monkeypatch.setattr(ops, "patch_scores", lambda *a, **k: (calls.__setitem__("patch", calls["patch"] + 1), op(*a, **k))[1])
    q, r = d["qry_cls"].to(dt), d["ref_cls"].to(dt)
    out = PairwiseSimilarity()(q.cuda(), r.cuda())
    assert calls["cos"] == 1 and out.dtype == torch.float32
    assert torch.allclose(out.cpu(), oism.pairwise_similarity(q.float(), r.float()), atol=2e-5)
    qp = d["qry_patch"].to(dt)
    ref = d["ref_patch"][d["gt_obj"], d["gt_tem"]].to(dt).contiguous()
    m = MaskedPatch_MatrixSimilarity()
    appe, ratio = m.both(qp.cuda(), ref.cuda(), 0.5)
    assert calls["patch"] == 1 and appe.dtype == dt
    want = oism.appearance_score(qp.float(), d["ref_patch"].to(dt).float(), d["gt_obj"], d["gt_tem"])[0]
    assert torch.allclose(appe.float().cpu(), want, atol=1e-2 if dt == torch.bfloat16 else 2e-3)       # output rounded to dt
    assert torch.allclose(ratio.float().cpu(), oism.visible_ratio(qp.float(), ref.float(), 0.5), atol=2e-2)


def test_batched_frames_equal_the_per_frame_loop():
    """FrameScorer.score_frames (F frames in one set of launches, per-mask frame index for the depth map / camera matrix, quirk
    Q3 applied per frame) == FrameScorer.score frame by frame, bit for bit."""
    from sam6d_amd.ism.scoring import FrameScorer
    F_, P = 3, 24
    frames = [synth.ism_inputs(P=P, O=2, T=6, C=128, n_patch=48, H=120, W=160, seed=20 + f) for f in range(F_)]
    frames[2]["K"] = frames[2]["K"].clone()
    frames[2]["K"][0, 0] *= 1.1                                          # a camera of its own
    base = frames[0]
    sc = FrameScorer(base["ref_cls"].cuda(), base["ref_patch"].cuda(), base["poses"].cuda(), base["pointcloud"].cuda(),
                     confidence_thresh=0.1)
    def one(fr):
        return sc.score(fr["qry_cls"].cuda(), fr["qry_patch"].cuda(), fr["masks"].cuda(), fr["boxes"].cuda(), fr["depth"].cuda(), fr["K"])
    # random boxes rarely all overlap their projections (quirk Q3 then zeroes the frame): give frames 0 and 2 boxes that do
    for f in (0, 2):
        r = one(frames[f])
        uv = r["image_uv"].cpu().float()
        bb = torch.cat((uv.min(1).values - 2, uv.max(1).values + 3), -1)
        frames[f]["boxes"] = frames[f]["boxes"].clone()
        frames[f]["boxes"][r["sel"].cpu()] = bb
    # frame 1: a far scene (small projections) and one selected proposal whose box sits in a corner: an empty intersection,
    # i.e. quirk Q3 zeroes the IoU of the whole frame
    frames[1]["depth"] = frames[1]["depth"] * 20
    r = one(frames[1])
    frames[1]["boxes"] = frames[1]["boxes"].clone()
    frames[1]["boxes"][r["sel"][0].item()] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    per = [one(fr) for fr in frames]
    st = lambda k: torch.stack([fr[k] for fr in frames]).cuda()
    out = sc.score_frames(st("qry_cls"), st("qry_patch"), st("masks"), st("boxes"), st("depth"), st("K"))
    assert any(not torch.is_tensor(p_["iou"]) for p_ in per) and any(torch.is_tensor(p_["iou"]) for p_ in per)   # both Q3 branches
    for f, ref in enumerate(per):
        rows = out["frame"] == f
        assert torch.equal(out["sel"][rows], ref["sel"])
        for k in ("pred_obj", "semantic", "best_template", "appearance", "visible_ratio", "final", "image_uv"):
            assert torch.equal(out[k][rows], ref[k]), (f, k)
        iou = ref["iou"] if torch.is_tensor(ref["iou"]) else torch.zeros_like(ref["final"])
        assert torch.equal(out["iou"][rows], iou)


def test_score_frames_with_no_selected_proposal():
    """a semantic threshold above every proposal's score -> empty outputs, no launch with S = 0."""
    from sam6d_amd.ism.scoring import FrameScorer
    fr = synth.ism_inputs(P=6, O=2, T=4, C=64, n_patch=16, H=120, W=160, seed=2)
    sc = FrameScorer(fr["ref_cls"].cuda(), fr["ref_patch"].cuda(), fr["poses"].cuda(), fr["pointcloud"].cuda(), confidence_thresh=2.0)
    st = lambda k: torch.stack([fr[k], fr[k]]).cuda()
    out = sc.score_frames(st("qry_cls"), st("qry_patch"), st("masks"), st("boxes"), st("depth"), st("K"))
    assert all(out[k].shape[0] == 0 for k in ("frame", "sel", "pred_obj", "semantic", "appearance", "iou", "final", "image_uv"))
