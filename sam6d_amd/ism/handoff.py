"""ISM -> PEM hand-off and result records.

The reference passes detections between its two models through files: per-frame ``.npz`` (Detections.save_to_file,
Instance_Segmentation_Model/model/utils.py:162-181), converted to one BOP-style JSON list whose masks are uncompressed
column-major RLE produced by a per-pixel Python loop (``mask_to_rle`` :24-43, ``convert_npz_to_json`` :199-216), which the
PEM data provider decodes again (Pose_Estimation_Model/utils/data_utils.py:72-89).  In one process per GPU the masks can
stay device tensors (``Detections`` below is that hand-off); when the unchanged eval scripts need the files, the same
records are produced from the tensors: the run lengths come from ONE pass of tensor ops over all masks (transitions ->
positions -> differences) instead of H*W Python iterations per mask, byte-identical with the reference's JSON.
"""
import json
from dataclasses import dataclass

import numpy as np
import torch

LMO_OBJECT_IDS = np.array([1, 5, 6, 8, 9, 10, 11, 12])                # model/utils.py:11-22


@dataclass
class Detections:
    """What the PEM stage needs from the ISM stage for one frame (the in-memory replacement of the JSON round trip):
    masks (N,H,W) bool, boxes (N,4) XYXY, scores (N,), object_ids (N,) zero-based, plus the frame identifiers."""
    scene_id: int
    image_id: int
    masks: torch.Tensor
    boxes: torch.Tensor
    scores: torch.Tensor
    object_ids: torch.Tensor
    runtime: float = 0.0

    # ---- the per-frame list operations of the reference's Detections (model/utils.py:80-132) -----------------------------
    _FIELDS = ("masks", "boxes", "scores", "object_ids")

    def __len__(self):
        return self.boxes.shape[0]

    def filter(self, idx):
        """Keep the rows ``idx`` (bool mask or index tensor) of every per-detection tensor (model/utils.py:filter; fields
        that are not filled yet -- scores / object_ids before the scoring stage -- are left alone)."""
        n = len(self)
        for f in self._FIELDS:
            v = getattr(self, f)
            if v is not None and v.shape[0] == n:
                setattr(self, f, v[idx])
        return self

    def remove_very_small_detections(self, min_box_size, min_mask_size):
        """model/utils.py:96-105: keep detections whose box covers more than min_box_size**2 of the frame AND whose mask
        covers more than min_mask_size of it (the asymmetry -- one threshold squared, the other not -- is the
        reference's; configs/model/ISM_sam.yaml:15-16 sets 0.05 and 3e-4)."""
        img_area = self.masks.shape[1] * self.masks.shape[2]
        b = self.boxes
        box_areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) / img_area          # torchvision box_area
        mask_areas = self.masks.sum(dim=(1, 2)) / img_area
        return self.filter(torch.logical_and(box_areas > min_box_size ** 2, mask_areas > min_mask_size))

    def apply_nms(self, nms_thresh=0.5, nms_fn=None):
        """model/utils.py:121-126."""
        nms_fn = nms_fn or _device_nms
        return self.filter(nms_fn(self.boxes.float(), self.scores.float(), nms_thresh))

    def apply_nms_per_object_id(self, nms_thresh=0.5, nms_fn=None):
        """model/utils.py:107-119: one NMS per predicted object, survivors concatenated in ascending object id, inside
        an object by decreasing score.  ``nms_fn(boxes, scores, thresh) -> kept indices by decreasing score``; default the
        device kernel (sam6d_amd.ops.nms -> s6d_nms_f32)."""
        nms_fn = nms_fn or _device_nms
        every = torch.arange(len(self), device=self.boxes.device)
        keep = []
        for oid in torch.unique(self.object_ids):
            sel = self.object_ids == oid
            keep.append(every[sel][nms_fn(self.boxes[sel].float(), self.scores[sel].float(), nms_thresh)])
        if not keep:
            return self                                               # nothing to do for an empty frame (the reference raises)
        return self.filter(torch.cat(keep))


def _device_nms(boxes, scores, thresh):
    from .. import ops
    return ops.nms(boxes.contiguous(), scores.contiguous(), thresh)


def masks_to_rle(masks):
    """(N,H,W) bool/0-1 tensor (any device) -> list of {"counts": [...], "size": [H, W]}: uncompressed COCO-style RLE in
    column-major order starting with the zero run (possibly 0 long), exactly model/utils.py:mask_to_rle."""
    m = (masks > 0) if masks.dtype != torch.bool else masks
    N, H, W = m.shape
    flat = m.transpose(1, 2).reshape(N, H * W)                          # column-major ("F") pixel order
    prev = torch.cat([torch.zeros(N, 1, dtype=torch.bool, device=m.device), flat[:, :-1]], dim=1)
    change = flat != prev                                               # a run starts here (virtual 0 before pixel 0)
    idx = torch.nonzero(change)                                         # sorted by mask, then position
    per_mask = torch.bincount(idx[:, 0], minlength=N).cpu().tolist()
    pos = idx[:, 1].cpu().numpy()
    out, a = [], 0
    for n in range(N):
        p = pos[a:a + per_mask[n]]
        a += per_mask[n]
        edges = np.concatenate([[0], p, [H * W]])
        out.append({"counts": np.diff(edges).tolist(), "size": [H, W]})
    return out


def rle_to_mask(rle):
    """Inverse (Pose_Estimation_Model/utils/data_utils.py:72-89), vectorised."""
    h, w = rle["size"]
    counts = np.asarray(rle["counts"], dtype=np.int64)
    vals = (np.arange(len(counts)) % 2).astype(bool)
    return np.repeat(vals, counts).reshape(h, w, order="F")


def detection_records(det: Detections, dataset_name):
    """The per-detection dicts of convert_npz_to_json (model/utils.py:199-216) for one frame, from device tensors:
    category ids shifted as in save_to_file :171-173, boxes XYXY -> XYWH without the +1 (the (N,4) branch of
    utils/bbox_utils.py:134-136)."""
    obj = det.object_ids.cpu().numpy()
    cat = LMO_OBJECT_IDS[obj] if dataset_name == "lmo" else obj + 1
    b = det.boxes.cpu().numpy()
    xywh = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1)
    rles = masks_to_rle(det.masks)
    scores = det.scores.cpu().numpy()
    return [{"scene_id": int(det.scene_id), "image_id": int(det.image_id), "category_id": int(cat[i]),
             "bbox": xywh[i].tolist(), "score": float(scores[i]), "time": float(det.runtime), "segmentation": rles[i]}
            for i in range(len(rles))]


def save_json_bop23(path, records):
    """utils/inout.py:62-65."""
    with open(path, "w") as f:
        json.dump(records, f)
