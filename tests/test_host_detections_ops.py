"""List operations on a frame's detections (the glue of detector.py:test_step around the scored path) against the
reference's Detections class and its segmentor's postprocess_resize run unmodified (golden: oracle/gen_golden.py
detections_ops; torchvision's nms / box_area supplied from their published definitions there)."""
import numpy as np
import torch

from oracle import sam_decoder as osd
from sam6d_amd.ism.handoff import Detections
from sam6d_amd.sam import amg
from tests import util


def inputs(g):
    N, H, W = (int(v) for v in g["shape"])
    masks = torch.from_numpy(np.unpackbits(g["masks"])[: N * H * W].reshape(N, H, W).astype(bool))
    return Detections(1, 2, masks, torch.from_numpy(g["boxes"]), torch.from_numpy(g["scores"]), torch.from_numpy(g["obj"]))


def test_remove_very_small_detections():
    g = util.golden("detections_ops.npz")
    d = inputs(g).remove_very_small_detections(0.05, 3e-2)
    np.testing.assert_array_equal(d.scores.numpy(), g["small_scores"])
    assert len(d) == len(g["small_scores"]) == d.masks.shape[0] == d.boxes.shape[0] == d.object_ids.shape[0]


def test_nms_and_nms_per_object_id_with_the_oracle_nms():
    g = util.golden("detections_ops.npz")
    d = inputs(g).apply_nms(0.5, nms_fn=osd.nms)
    np.testing.assert_array_equal(d.scores.numpy(), g["nms_scores"])
    d = inputs(g).apply_nms_per_object_id(0.25, nms_fn=osd.nms)
    np.testing.assert_array_equal(d.scores.numpy(), g["nms_obj_scores"])
    np.testing.assert_array_equal(d.object_ids.numpy(), g["nms_obj_ids"])
    np.testing.assert_array_equal(d.boxes.numpy(), g["nms_obj_boxes"])
    np.testing.assert_array_equal(d.masks.sum(dim=(1, 2)).numpy(), g["nms_obj_mask_sums"])
    assert (np.diff(g["nms_obj_ids"]) >= 0).all()                   # ascending object id, the reference's order
    empty = Detections(0, 0, torch.zeros(0, 4, 4, dtype=torch.bool), torch.zeros(0, 4, dtype=torch.long), torch.zeros(0),
                       torch.zeros(0, dtype=torch.long))
    assert len(empty.apply_nms_per_object_id(0.25, nms_fn=osd.nms)) == 0


def test_filter_before_scoring_leaves_unfilled_fields():
    g = util.golden("detections_ops.npz")
    d = inputs(g)
    d.scores, d.object_ids = None, None
    d.filter(torch.tensor([3, 1]))
    assert len(d) == 2 and d.scores is None and torch.equal(d.boxes, torch.from_numpy(g["boxes"])[[3, 1]])


def test_postprocess_resize_matches_the_reference():
    g = util.golden("detections_ops.npz")
    d = inputs(g)
    for tag, orig in (("same", (48, 64)), ("up", (81, 108))):
        m, b = amg.postprocess_resize(d.masks[:6], d.boxes[:6], orig, 64)
        np.testing.assert_array_equal(m.numpy(), g["pp_masks_" + tag])
        np.testing.assert_array_equal(b.numpy(), g["pp_boxes_" + tag])
    assert m.dtype == torch.float32 and b.dtype == torch.float32
    assert amg.segmentor_input_size((480, 640), 640) == (480, 640) and amg.segmentor_input_size((540, 720), 640) == (480, 640)
    assert amg.segmentor_input_size((960, 1280), 640) == (480, 640) and amg.segmentor_input_size((100, 333), 640) == (192, 640)
