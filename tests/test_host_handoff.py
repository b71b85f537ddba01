"""ISM -> PEM hand-off records against the reference's own writers: RLE of every mask and the
per-detection JSON records are byte-identical (golden made by model/utils.py functions run unmodified)."""
import json

import numpy as np
import torch

from sam6d_amd.ism import handoff
from tests import util


def _inputs():
    g = torch.Generator().manual_seed(12)
    H, W, N = 37, 53, 6
    masks = (torch.rand(N, H, W, generator=g) > 0.55)
    masks[0] = False
    masks[1] = True
    masks[2, :, :7] = True
    boxes = torch.tensor([[0, 0, 0, 0], [0, 0, W - 1, H - 1], [0, 0, 6, H - 1], [3, 4, 30, 20], [10, 2, 50, 36], [1, 1, 2, 2]])
    scores = torch.rand(N, generator=g)
    obj = torch.tensor([0, 1, 2, 3, 7, 5])
    return masks, boxes, scores, obj


def test_rle_matches_reference_and_round_trips():
    g = util.golden("handoff.npz")
    masks, _, _, _ = _inputs()
    rles = handoff.masks_to_rle(masks)
    assert json.dumps(rles) == str(g["rle_json"])
    for m, r in zip(masks, rles):
        assert sum(r["counts"]) == m.numel() and np.array_equal(handoff.rle_to_mask(r), m.numpy())
    assert handoff.masks_to_rle(masks[:0]) == []
    assert handoff.masks_to_rle(masks.float()) == rles                       # 0/1 float masks as well


def test_detection_records_match_reference_json():
    g = util.golden("handoff.npz")
    masks, boxes, scores, obj = _inputs()
    for ds in ("lmo", "ycbv"):
        det = handoff.Detections(scene_id=2, image_id=17, masks=masks, boxes=boxes, scores=scores, object_ids=obj, runtime=0.25)
        assert json.dumps(handoff.detection_records(det, ds)) == str(g[ds + "_json"])
