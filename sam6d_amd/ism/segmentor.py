"""The ISM's segmentor plugin on MI355X: ``CustomSamAutomaticMaskGenerator`` (Instance_Segmentation_Model/model/sam.py:50-148,
Hydra ``_target_`` of configs/model/segmentor_model/sam.yaml) and ``load_sam`` (:22-37).

``generate_masks(image)`` is the whole per-frame proposal stage: optional resize of the frame to ``segmentor_width_size``,
SAM's own resize + normalisation + padding, the ViT image encoder, the 32 x 32 point-prompt grid through prompt encoder and
mask decoder in batches, the IoU / stability filters, box NMS, and the resize of masks and boxes back to the frame.  In
the reference that is SamPredictor + SamAutomaticMaskGenerator with an RLE encode / decode round trip per mask on the
host; here the frame is uploaded once and everything stays on the device (sam6d_amd.sam.amg.generate_proposals and the
kernels behind it).  Same constructor arguments, same return value: ``{"masks": (N,H,W), "boxes": (N,4)}`` -- bool masks
and integer XYXY boxes without ``segmentor_width_size``, float masks (bilinear, not re-thresholded) and float boxes
with it, exactly as the reference returns them.

Not provided: ``min_mask_region_area > 0`` (postprocess_small_regions: cv2 connected components, an un-vendored dependency;
the ISM configuration sets 0) and crop layers (``crop_n_layers`` is 0 in the reference's generator as the ISM builds it).
When the frame is not ``segmentor_width_size`` wide the reference shrinks it with ``cv2.resize``; here Pillow's bilinear
resampler (sam/transforms.py) does it -- same filter, possibly different rounding; for 640-wide frames it is the identity.
"""
import os

import numpy as np
import torch

from ..sam import amg
from ..sam.build_sam import sam_model_registry
from ..sam.transforms import ResizeLongestSide, pil_bilinear_resize_u8

pretrained_weight_dict = {"vit_h": "sam_vit_h_4b8939.pth", "vit_l": "sam_vit_l_0b3195.pth", "vit_b": "sam_vit_b_01ec64.pth"}   # model/sam.py:15-19


def load_sam(model_type, checkpoint_dir, device=None):
    """model/sam.py:22-27 (the reference leaves the move to the device to its caller; pass ``device`` to do it here)."""
    sam = sam_model_registry[model_type](checkpoint=os.path.join(checkpoint_dir, pretrained_weight_dict[model_type]))
    return sam if device is None else sam.to(device)


class CustomSamAutomaticMaskGenerator:
    def __init__(self, sam, min_mask_region_area=0, points_per_batch=64, stability_score_thresh=0.85, box_nms_thresh=0.7,
                 crop_overlap_ratio=512 / 1500, segmentor_width_size=None, pred_iou_thresh=0.88):
        if min_mask_region_area > 0:
            raise NotImplementedError("min_mask_region_area > 0 needs cv2 connected components (postprocess_small_regions); "
                                      "the ISM configuration uses 0")
        self.sam = sam
        self.points_per_batch = points_per_batch
        self.stability_score_thresh = stability_score_thresh
        self.box_nms_thresh = box_nms_thresh
        self.crop_overlap_ratio = crop_overlap_ratio                 # unused with crop_n_layers = 0, kept for the config
        self.segmentor_width_size = segmentor_width_size
        self.pred_iou_thresh = pred_iou_thresh
        # SamAutomaticMaskGenerator defaults the reference does not override (automatic_mask_generator.py:36-51)
        self.points_per_side, self.stability_score_offset, self.min_mask_region_area = 32, 1.0, 0

    @torch.no_grad()
    def generate_masks(self, image):
        """image: HxWx3 uint8 RGB (numpy as in the reference, or a tensor already on the device)."""
        sam = self.sam
        frame = torch.from_numpy(np.ascontiguousarray(image)) if isinstance(image, np.ndarray) else image
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise ValueError("expected an HxWx3 uint8 image")
        frame = frame.to(sam.device)
        orig_size = tuple(frame.shape[:2])
        if self.segmentor_width_size is not None:
            frame = pil_bilinear_resize_u8(frame, amg.segmentor_input_size(orig_size, self.segmentor_width_size))
        size = tuple(frame.shape[:2])
        enc = sam.image_encoder
        x = ResizeLongestSide(enc.img_size).apply_image(frame).permute(2, 0, 1)[None].float()
        emb = enc(sam.preprocess(x)).float()
        prop = amg.generate_proposals(sam.prompt_encoder, sam.mask_decoder, emb, size, enc.img_size,
                                      points_per_side=self.points_per_side, points_per_batch=self.points_per_batch,
                                      mask_threshold=sam.mask_threshold, pred_iou_thresh=self.pred_iou_thresh,
                                      stability_score_thresh=self.stability_score_thresh,
                                      stability_score_offset=self.stability_score_offset, box_nms_thresh=self.box_nms_thresh)
        masks, boxes = prop["masks"], prop["boxes"]
        if self.segmentor_width_size is not None:
            masks, boxes = amg.postprocess_resize(masks, boxes, orig_size, self.segmentor_width_size)
        return {"masks": masks, "boxes": boxes}
