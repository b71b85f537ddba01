"""A pixels-to-proposals case in which the segmentor's box NMS really suppresses (VERDICT r4 missing #3 / next #1e).

tests/golden/frame_e2e_nms.npz (oracle/gen_golden.py frame_e2e_nms): the reference's CustomSamAutomaticMaskGenerator on the Example
frame with the seeded SAM and ``Sam.mask_threshold = 0.18`` -- above the 99.9th percentile of the seeded mask logits, so a mask is a
sparse set of its own highest pixels and its box follows the prompt -- at the reference's own box_nms_thresh = 0.7: 94 candidates
reach batched_nms (model/sam.py:138-144), 11 survive.  (In frame_e2e.npz every box is the frame and the threshold is 1.5.)
  * the product's NMS kernel (sam6d_amd.ops.nms -> s6d_nms_f32) on the 94 boxes / scores the reference's call received keeps
    exactly the reference's 11, in its order;
  * the product's generator, free-running from the pixels in float32, returns the same proposals: boxes identical for at least
    nine of the eleven (a candidate whose stability score or predicted IoU sits within float32 noise of its threshold may enter
    or leave the set of 94, and with it one suppression chain), masks of the matched proposals identical up to 2 pixels."""
import ast

import numpy as np
import pytest
import torch

from sam6d_amd.utils import seeded
from tests import util

pytestmark = pytest.mark.gpu


def test_nms_kernel_on_the_reference_generators_own_candidates():
    from sam6d_amd import ops
    g = util.golden("frame_e2e_nms.npz")
    boxes, scores = torch.from_numpy(g["nms_in_boxes"]).float().cuda(), torch.from_numpy(g["nms_in_scores"]).float().cuda()
    c = ast.literal_eval(str(g["case"]))
    keep = ops.nms(boxes.contiguous(), scores.contiguous(), c["box_nms_thresh"])
    assert len(g["nms_in_scores"]) > 5 * len(g["nms_keep"]) > 0, "the case is meant to suppress most of its candidates"
    assert keep.cpu().tolist() == g["nms_keep"].tolist()


def test_generator_free_running_returns_the_reference_proposals(monkeypatch):
    from sam6d_amd.ism.segmentor import CustomSamAutomaticMaskGenerator
    from sam6d_amd.sam.build_sam import sam_model_registry
    g = util.golden("frame_e2e_nms.npz")
    c = ast.literal_eval(str(g["case"]))
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "fp32")
    monkeypatch.setenv("S6D_SAM_DTYPE", "fp32")
    sam = seeded.load_seeded(sam_model_registry["vit_h"]().eval(), c["sam_seed"]).cuda()
    sam.mask_threshold = c["mask_threshold"]
    gen = CustomSamAutomaticMaskGenerator(sam, points_per_batch=256, stability_score_thresh=c["stability_score_thresh"],
                                          pred_iou_thresh=c["pred_iou_thresh"], box_nms_thresh=c["box_nms_thresh"])
    gen.stability_score_offset = c["stability_score_offset"]
    rgb = util.golden("example_frame.npz")["rgb"]
    det = gen.generate_masks(rgb)
    K = g["boxes"].shape[0]
    want_boxes = g["boxes"].astype(np.int64)
    want_masks = np.unpackbits(g["masks"], axis=1)[:, :480 * 640].reshape(K, 480, 640).astype(bool)
    got_boxes, got_masks = det["boxes"].cpu().numpy().astype(np.int64), det["masks"].cpu().numpy()
    matched, worst = 0, 0
    for i in range(K):
        j = np.nonzero((got_boxes == want_boxes[i]).all(1))[0]
        if len(j):
            matched += 1
            worst = max(worst, int((got_masks[j[0]] != want_masks[i]).sum()))
    util.record_margin("frame_e2e_nms", reference_proposals=K, product_proposals=len(got_boxes), boxes_identical=matched, mask_pixels_differing_max=worst)
    assert abs(len(got_boxes) - K) <= 2 and matched >= K - 2, (got_boxes.tolist(), want_boxes.tolist())
    assert worst <= 2
