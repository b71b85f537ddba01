"""Per-batch body of SAM's automatic mask generator on MI355X.

``process_point_batch`` is what ``SamAutomaticMaskGenerator._process_batch`` (segment_anything/
automatic_mask_generator.py:266-312) computes between "run the model on this batch of points" and "compress to RLE":
prompt encoder -> mask decoder -> upscaling to the frame -> predicted-IoU filter -> stability score + filter ->
threshold -> boxes.  The reference goes through ``SamPredictor.predict_torch(return_logits=True)`` (predictor.py:169-243),
i.e. it materialises the (64*3, H, W) fp32 logits of every batch; here the decoder's 256 x 256 logits go straight into
``s6d_sam_mask_post_f32`` and only binary masks, two counts and a box per mask exist at frame resolution.  The filters
commute (each is a per-mask predicate), so they are applied together after the one fused pass.
"""
import torch

from .. import ops


@torch.no_grad()
def process_point_batch(prompt_encoder, mask_decoder, image_embedding, in_points, input_size, original_size, img_size=1024,
                        mask_threshold=0.0, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                        stability_score_offset=1.0):
    """image_embedding (1,C,h,w) of the frame; in_points (B,2) point prompts in the resized input frame (what
    ``ResizeLongestSide.apply_coords`` returns, automatic_mask_generator.py:276-277); input_size = (h,w) of the resized
    frame inside the padded img_size square; original_size = (H,W) of the frame.
    -> dict(masks bool (K,H,W), iou_preds (K,), stability_score (K,), boxes (K,4) long XYXY, point_index (K,) long):
    the masks that pass both filters, in the reference's (prompt-major, then the 3 multimask outputs) order."""
    B = in_points.shape[0]
    labels = torch.ones(B, 1, dtype=torch.int, device=in_points.device)
    sparse, dense = prompt_encoder(points=(in_points[:, None, :], labels), boxes=None, masks=None)
    low_res, iou = mask_decoder(image_embeddings=image_embedding, image_pe=prompt_encoder.get_dense_pe(),
                                sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, multimask_output=True)
    masks, stability, boxes = ops.sam_mask_post(low_res.float().contiguous(), img_size, input_size, original_size,
                                                mask_threshold, stability_score_offset)
    iou = iou.flatten(0, 1)
    keep = torch.ones_like(iou, dtype=torch.bool)
    if pred_iou_thresh > 0.0:
        keep &= iou > pred_iou_thresh
    if stability_score_thresh > 0.0:
        keep &= stability >= stability_score_thresh                  # NaN (empty +-offset masks) fails, as in the reference
    idx = torch.nonzero(keep).squeeze(1)
    C = low_res.shape[1]
    return dict(masks=masks[idx], iou_preds=iou[idx], stability_score=stability[idx], boxes=boxes[idx],
                point_index=idx // C, low_res_logits=low_res)
