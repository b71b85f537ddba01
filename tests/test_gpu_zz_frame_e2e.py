"""The Example frame from PIXELS to scored detections against tests/golden/frame_e2e.npz: the golden was
made by the reference's own SamPredictor + CustomSamAutomaticMaskGenerator + CustomDINOv2 + scoring methods (oracle/gen_golden.py
frame_e2e, seeded weights); here the PRODUCT's drop-in modules run the same chain on the device.

  * fp32 chain (every model in float32 on the device): the proposal set, its boxes and the integer decisions of the scoring stage
    (sel, pred_obj, best_template) must be the reference's; masks agree up to the pixels whose logit sits within float32 noise of
    the threshold (counted and bounded), descriptors to 1e-3, scores to 1e-4.
  * benched dtypes (bf16 SAM encoder / mask decoder / DINOv2): DECISION-level comparison with the fp32 chain -- candidate masks'
    IoU, filter-decision flips, and flips of sel / pred_obj / best_template on the same proposals -- with stated bounds."""
import ast
import os

import numpy as np
import pytest
import torch

from sam6d_amd.utils import seeded, synth
from tests import util

pytestmark = pytest.mark.gpu


def _case():
    g = util.golden("frame_e2e.npz")
    return g, ast.literal_eval(str(g["case"]))


def _extra(c):
    """The model-independent inputs of the case (oracle/gen_golden.py::e2e_templates_and_extra, restated: the GPU box has no oracle
    import restrictions in tests, but the generator module needs /root/reference at import of its harness only lazily -- the two
    helpers it uses are importable anywhere)."""
    fi = util.frame_inputs(dict(P=10, O=1, T=6, C=128, n_patch=64, seed=21))
    poses = synth.ism_inputs(P=4, O=c["O"], T=c["T"], C=8, n_patch=4, H=480, W=640, seed=c["ism_seed"])["poses"]
    return fi, poses


def _segmentor(c, dtype):
    from sam6d_amd.ism.segmentor import CustomSamAutomaticMaskGenerator
    from sam6d_amd.sam.build_sam import sam_model_registry
    sam = seeded.load_seeded(sam_model_registry["vit_h"]().eval(), c["sam_seed"]).cuda()
    if dtype == torch.bfloat16:
        sam.image_encoder.to(torch.bfloat16)
    gen = CustomSamAutomaticMaskGenerator(sam, points_per_batch=256, stability_score_thresh=c["stability_score_thresh"],
                                          pred_iou_thresh=c["pred_iou_thresh"], box_nms_thresh=c["box_nms_thresh"])
    gen.stability_score_offset = c["stability_score_offset"]
    return sam, gen


def _descriptor_model(c):
    from sam6d_amd.ism import dinov2 as pd
    o = pd.CustomDINOv2.__new__(pd.CustomDINOv2)
    torch.nn.Module.__init__(o)
    o.model = seeded.load_seeded(pd._make_dinov2_model(arch_name="vit_large").eval(), c["dino_seed"]).cuda()
    o.patch_size, o.validpatch_thresh, o.chunk_size, o.proposal_size, o.token_name = 14, 0.5, 64, 224, "x_norm_clstoken"
    return o


def _scorer(g, c, poses, fi):
    from sam6d_amd.ism.scoring import FrameScorer
    pointcloud = fi["pointcloud"] * torch.tensor([1.0, 0.8, 1.2])[:c["O"]].view(-1, 1, 1)           # one model cloud per object
    return FrameScorer(torch.from_numpy(g["ref_cls"]).float().cuda(), torch.from_numpy(g["ref_patch"]).float().cuda(), poses.cuda(),
                       pointcloud.cuda(), confidence_thresh=c["confidence_thresh"])


def _score(fs, cls, patch, masks, boxes, fi):
    return fs.score(cls, patch, masks.float(), boxes.float(), fi["depth_mm"].cuda(), fi["K"], depth_scale=fi["depth_scale"])


def test_frame_e2e_fp32_chain_vs_reference_golden(monkeypatch):
    from types import SimpleNamespace
    g, c = _case()
    fi, poses = _extra(c)
    rgb = fi["rgb"]
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "fp32")
    monkeypatch.setenv("S6D_SAM_DTYPE", "fp32")
    monkeypatch.setenv("S6D_DINO_DTYPE", "fp32")
    sam, gen = _segmentor(c, torch.float32)
    det = gen.generate_masks(rgb)
    K = g["sam_boxes"].shape[0]
    want = torch.from_numpy(np.unpackbits(g["sam_masks"], axis=1)[:, :480 * 640].reshape(K, 480, 640).astype(bool))
    assert det["masks"].shape[0] == K, (det["masks"].shape, K)
    got = det["masks"].cpu()
    diff = (got != want).flatten(1).sum(1)
    union = (got | want).flatten(1).sum(1).clamp(min=1)
    util.record_margin("frame_e2e_fp32_sam", proposals=K, pixels_differing_max=int(diff.max()), pixels_differing_total=int(diff.sum()),
                       iou_min=float(1 - (diff / union).max()))
    # float32 on two machines: a pixel whose logit sits within rounding noise of the threshold may land on either side
    assert (diff / union).max() < 1e-3, diff.tolist()
    dbox = (det["boxes"].cpu() - torch.from_numpy(g["sam_boxes"])).abs()
    assert (dbox.max(1).values == 0).float().mean() >= 0.9 and dbox.max() <= 2, dbox.max(1).values.tolist()
    del sam, gen
    torch.cuda.empty_cache()
    # ---- descriptors of SAM's proposals + the ten depth-window proposals, scoring -----------------------------------------------
    masks = torch.cat([want.float(), fi["masks"]]).cuda()                    # the reference's proposals: stage-wise comparison
    boxes = torch.cat([torch.from_numpy(g["sam_boxes"]).float(), fi["boxes"]]).cuda()
    o = _descriptor_model(c)
    cls, patch = o.forward(rgb, SimpleNamespace(masks=masks, boxes=boxes))
    ref = torch.from_numpy(g["qry_cls"])
    rel = ((cls.cpu() - ref).norm(dim=1) / ref.norm(dim=1)).max().item()
    util.record_margin("frame_e2e_fp32_descriptors", cls_rel_max=rel)
    assert rel < 1e-3, rel
    util.assert_digest_close(patch.cpu(), g["qry_patch_sum"], g["qry_patch_smp"], 211, 1e-3, 1e-4, "patch descriptors")
    sc = _score(_scorer(g, c, poses, fi), cls, patch, masks, boxes, fi)
    assert sc["sel"].cpu().tolist() == g["sel"].tolist()
    assert sc["pred_obj"].cpu().tolist() == g["pred_obj"].tolist()
    assert sc["best_template"].cpu().tolist() == g["best_template"].tolist()
    for k in ("semantic", "appearance", "visible_ratio", "final"):
        np.testing.assert_allclose(sc[k].cpu().numpy(), g[k], rtol=0, atol=1e-4, err_msg=k)
    assert np.array_equal(sc["image_uv"].cpu().numpy(), g["image_uv"])
    # ---- free-running: the product's own proposals instead of the reference's give the same decisions --------------------------
    masks2 = torch.cat([got.float(), fi["masks"]]).cuda()
    boxes2 = torch.cat([det["boxes"].cpu().float(), fi["boxes"]]).cuda()
    cls2, patch2 = o.forward(rgb, SimpleNamespace(masks=masks2, boxes=boxes2))
    sc2 = _score(_scorer(g, c, poses, fi), cls2, patch2, masks2, boxes2, fi)
    assert sc2["sel"].cpu().tolist() == g["sel"].tolist() and sc2["pred_obj"].cpu().tolist() == g["pred_obj"].tolist()
    assert sc2["best_template"].cpu().tolist() == g["best_template"].tolist()
    np.testing.assert_allclose(sc2["final"].cpu().numpy(), g["final"], rtol=0, atol=2e-3)


def test_frame_e2e_benched_dtypes_decisions_vs_fp32_chain(monkeypatch):
    """bf16 (what bench.py and the whole-frame pipeline run) against the fp32 chain of the same modules on the same frame, at the
    level of DECISIONS.  Bounds are stated next to each assertion; measured values go to the margins file."""
    from types import SimpleNamespace

    from sam6d_amd.sam import amg
    from sam6d_amd.sam.image_encoder import preprocess
    from sam6d_amd.sam.transforms import ResizeLongestSide
    g, c = _case()
    fi, poses = _extra(c)
    rgb = fi["rgb"]
    frame = torch.from_numpy(np.ascontiguousarray(rgb)).cuda()
    res = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", name)
        monkeypatch.setenv("S6D_SAM_DTYPE", name)
        sam, gen = _segmentor(c, dt)
        enc = sam.image_encoder
        x = ResizeLongestSide(enc.img_size).apply_image(frame).permute(2, 0, 1)[None].float()
        with torch.no_grad():
            emb = enc(sam.preprocess(x)).float()
            parts = []
            pts = torch.as_tensor(amg.build_point_grid(32) * [[640, 480]] * np.array([[1024 / 640, 768 / 480]]), device="cuda")
            for a in range(0, 1024, 256):                                 # every candidate: no filter (thresholds <= 0 switch them off)
                r = amg.process_point_batch(sam.prompt_encoder, sam.mask_decoder, emb, pts[a:a + 256], (768, 1024), (480, 640), 1024,
                                            sam.mask_threshold, 0.0, 0.0, c["stability_score_offset"])
                parts.append({k: r[k] for k in ("masks", "iou_preds", "stability_score")})
        res[name] = dict(emb=emb, masks=torch.cat([p["masks"] for p in parts]), iou=torch.cat([p["iou_preds"] for p in parts]).float(),
                         stab=torch.cat([p["stability_score"] for p in parts]).float())
        assert res[name]["masks"].shape[0] == 3072
        del sam, gen
        torch.cuda.empty_cache()
    a, b = res["fp32"], res["bf16"]
    emb_rel = ((a["emb"] - b["emb"]).pow(2).mean().sqrt() / a["emb"].pow(2).mean().sqrt()).item()
    inter = (a["masks"] & b["masks"]).flatten(1).sum(1).float()
    union = (a["masks"] | b["masks"]).flatten(1).sum(1).float().clamp(min=1)
    miou = inter / union
    keep_a = (a["iou"] > c["pred_iou_thresh"]) & (a["stab"] >= c["stability_score_thresh"])
    keep_b = (b["iou"] > c["pred_iou_thresh"]) & (b["stab"] >= c["stability_score_thresh"])
    flips = (keep_a != keep_b).float().mean().item()
    util.record_margin("frame_e2e_bf16_vs_fp32_sam", emb_rel=emb_rel, mask_iou_mean=miou.mean().item(), mask_iou_min=miou.min().item(),
                       mask_iou_p05=torch.quantile(miou, 0.05).item(), iou_pred_diff_max=(a["iou"] - b["iou"]).abs().max().item(),
                       stability_diff_max=(a["stab"] - b["stab"]).abs().max().item(), filter_flip_rate=flips,
                       kept_fp32=int(keep_a.sum()), kept_bf16=int(keep_b.sum()))
    # seeded weights give logits of +-0.05 around the threshold everywhere (a trained decoder's are +-10): the mask IoU under bf16
    # rounding is the hardest case there is.  Measured in round 4 (profiles/r04_parity_margins_final.jsonl): embedding 1.4e-2 of
    # its rms off, mask IoU mean 0.991 / minimum 0.970 over the 3072 candidates, predicted IoU within 2.8e-3, stability score
    # within 7.8e-3, ONE of the 3072 filter decisions flipped (3.3e-4).  Bounds = about twice the measured distance.
    assert emb_rel < 2.5e-2, emb_rel                       # the encoder's error model: 4e-3 * sqrt(33) (tests/test_gpu_sam.py)
    assert miou.mean() > 0.98 and miou.min() > 0.94, (miou.mean().item(), miou.min().item())
    assert (a["iou"] - b["iou"]).abs().max() < 6e-3 and (a["stab"] - b["stab"]).abs().max() < 1.6e-2
    assert flips <= 3 / 3072 + 1e-9, flips
    del res, a, b
    torch.cuda.empty_cache()
    # ---- descriptors + scoring on the SAME proposals (the golden's), bf16 against fp32 -------------------------------------------
    K = g["sam_boxes"].shape[0]
    want = torch.from_numpy(np.unpackbits(g["sam_masks"], axis=1)[:, :480 * 640].reshape(K, 480, 640).astype(bool))
    masks = torch.cat([want.float(), fi["masks"]]).cuda()
    boxes = torch.cat([torch.from_numpy(g["sam_boxes"]).float(), fi["boxes"]]).cuda()
    o = _descriptor_model(c)
    out = {}
    for name in ("fp32", "bf16"):
        monkeypatch.setenv("S6D_DINO_DTYPE", name)
        cls, patch = o.forward(rgb, SimpleNamespace(masks=masks, boxes=boxes))
        out[name] = (cls.float(), _score(_scorer(g, c, poses, fi), cls.float(), patch.float(), masks, boxes, fi))
    c32, s32 = out["fp32"]
    c16, s16 = out["bf16"]
    cos = torch.nn.functional.cosine_similarity(c32, c16, dim=1)
    same_sel = s32["sel"].tolist() == s16["sel"].tolist()
    n = min(len(s32["sel"]), len(s16["sel"]))
    obj_flips = (s32["pred_obj"][:n] != s16["pred_obj"][:n]).float().mean().item() if same_sel else 1.0
    tpl_flips = (s32["best_template"][:n] != s16["best_template"][:n]).float().mean().item() if same_sel else 1.0
    dfinal = (s32["final"][:n] - s16["final"][:n]).abs().max().item() if same_sel else float("nan")
    util.record_margin("frame_e2e_bf16_vs_fp32_scoring", cls_cos_min=cos.min().item(), same_sel=same_sel, pred_obj_flip_rate=obj_flips,
                       best_template_flip_rate=tpl_flips, final_score_diff_max=dfinal)
    # measured in round 4: cosine >= 0.99993, the same 26 proposals selected, no object and no template flipped, final scores
    # within 1.9e-4.  The selection and the object decision must not move; one template of 26 may (near-ties between template views)
    assert cos.min() > 0.9998, cos.min().item()
    assert same_sel and obj_flips == 0.0, (s32["sel"].tolist(), s16["sel"].tolist(), obj_flips)
    assert tpl_flips <= 1 / 26 + 1e-9 and dfinal < 1e-3, (tpl_flips, dfinal)
