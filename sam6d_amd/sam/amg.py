"""SAM's automatic mask generator on MI355X: the per-batch body and the single-crop generator.

``process_point_batch`` is what ``SamAutomaticMaskGenerator._process_batch`` (segment_anything/
automatic_mask_generator.py:266-312) computes between "run the model on this batch of points" and "compress to RLE":
prompt encoder -> mask decoder -> upscaling to the frame -> predicted-IoU filter -> stability score + filter ->
threshold -> boxes.  The reference goes through ``SamPredictor.predict_torch(return_logits=True)`` (predictor.py:169-243),
i.e. it materialises the (64*3, H, W) fp32 logits of every batch; here the decoder's 256 x 256 logits go straight into
``s6d_sam_mask_post_f32`` and only binary masks, two counts and a box per mask exist at frame resolution.  The filters
commute (each is a per-mask predicate), so they are applied together after the one fused pass.
"""
import os
import time

import numpy as np
import torch

from .. import ops
from .. import policy


def _decode_batch(prompt_encoder, mask_decoder, image_embedding, in_points, input_size, original_size, img_size, mask_threshold,
                  stability_score_offset):
    """prompt encoder -> mask decoder -> fused post-processing for one batch of point prompts -> (low_res, iou (B,C), masks
    (B*C,H,W) bool, stability (B*C,), boxes (B*C,4))."""
    B = in_points.shape[0]
    labels = torch.ones(B, 1, dtype=torch.int, device=in_points.device)
    sparse, dense = prompt_encoder(points=(in_points[:, None, :], labels), boxes=None, masks=None)
    low_res, iou = mask_decoder(image_embeddings=image_embedding, image_pe=prompt_encoder.get_dense_pe(),
                                sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, multimask_output=True)
    masks, stability, boxes = ops.sam_mask_post(low_res.float(), img_size, input_size, original_size, mask_threshold,
                                                stability_score_offset)
    return low_res, iou, masks, stability, boxes


_GRAPHS = {}          # (modules, shapes, dtypes, thresholds, weight fingerprint) -> (graph, static inputs, static outputs)
_GRAPH_MAX = 4        # distinct configurations kept (a segmentor serves one frame size: one entry)


def _weights_key(*modules):
    ver, ptr = 0, 0
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            ver += t._version
            ptr ^= t.data_ptr()
    return ver, ptr


def _decode_batch_graphed(prompt_encoder, mask_decoder, image_embedding, in_points, input_size, original_size, img_size,
                          mask_threshold, stability_score_offset):
    """``_decode_batch`` replayed as a hipGraph (S6D_AMG_GRAPH=0 turns it off).  The batch is ~150 launches, most of them the small
    token-side operations of the two-way transformer (7 tokens per prompt): measured 19.6 ms per 1024 prompts eager against ~13 ms
    of kernel time (profiles/r04_frame_demo_kernel_stats_first.csv) -- the rest is the host issuing them.  Nothing inside depends on
    a value read back from the device; the filters behind it (torch.nonzero) stay outside the graph.  Static input buffers
    (embedding, prompts) are overwritten before each replay; the outputs are the graph's own buffers, valid until the next call
    with the same configuration (the caller gathers what it keeps right away).  The key carries everything a capture bakes in."""
    dev = image_embedding.device
    key = (id(prompt_encoder), id(mask_decoder), tuple(image_embedding.shape), image_embedding.dtype, tuple(in_points.shape),
           in_points.dtype, tuple(input_size), tuple(original_size), int(img_size), float(mask_threshold), float(stability_score_offset),
           policy.version(),
           torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype(),
           _weights_key(prompt_encoder, mask_decoder), dev.index)
    g = _GRAPHS.get(key)
    if g is None:
        if len(_GRAPHS) >= _GRAPH_MAX:
            _GRAPHS.clear()
        emb_s, pts_s = image_embedding.clone(), in_points.clone()
        args = (prompt_encoder, mask_decoder, emb_s, pts_s, input_size, original_size, img_size, mask_threshold, stability_score_offset)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                                      # warm every lazily filled cache of the decoder
                _decode_batch(*args)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = _decode_batch(*args)
        g = (graph, emb_s, pts_s, outs)
        _GRAPHS[key] = g
    graph, emb_s, pts_s, outs = g
    emb_s.copy_(image_embedding)
    pts_s.copy_(in_points)
    graph.replay()
    return outs


def invalidate_graphs():
    """Drop every captured decoder graph (re-captured on the next call)."""
    _GRAPHS.clear()


@torch.no_grad()
def process_point_batch(prompt_encoder, mask_decoder, image_embedding, in_points, input_size, original_size, img_size=1024,
                        mask_threshold=0.0, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                        stability_score_offset=1.0):
    """image_embedding (1,C,h,w) of the frame; in_points (B,2) point prompts in the resized input frame (what
    ``ResizeLongestSide.apply_coords`` returns, automatic_mask_generator.py:276-277); input_size = (h,w) of the resized
    frame inside the padded img_size square; original_size = (H,W) of the frame.
    -> dict(masks bool (K,H,W), iou_preds (K,), stability_score (K,), boxes (K,4) long XYXY, point_index (K,) long):
    the masks that pass both filters, in the reference's (prompt-major, then the 3 multimask outputs) order."""
    graphed = (torch.cuda.is_available() and image_embedding.is_cuda and policy.current().amg_graph == "1"
               and in_points.shape[0] >= 64 and not torch.cuda.is_current_stream_capturing())
    fn = _decode_batch_graphed if graphed else _decode_batch
    low_res, iou, masks, stability, boxes = fn(prompt_encoder, mask_decoder, image_embedding, in_points, input_size, original_size,
                                               img_size, mask_threshold, stability_score_offset)
    iou = iou.flatten(0, 1)
    keep = torch.ones_like(iou, dtype=torch.bool)
    if pred_iou_thresh > 0.0:
        keep &= iou > pred_iou_thresh
    if stability_score_thresh > 0.0:
        keep &= stability >= stability_score_thresh                  # NaN (empty +-offset masks) fails, as in the reference
    idx = torch.nonzero(keep).squeeze(1)
    C = low_res.shape[1]
    # low_res of the graphed path is the graph's own output buffer, overwritten by the next replay: callers that keep several
    # batches (generate_proposals) get a private copy
    return dict(masks=masks[idx], iou_preds=iou[idx], stability_score=stability[idx], boxes=boxes[idx],
                point_index=idx // C, low_res_logits=low_res.clone() if graphed else low_res)


def build_point_grid(n_per_side):
    """utils/amg.py:179-186: n x n points at the pixel-cell centres of the unit square, (x, y), row-major."""
    offset = 1 / (2 * n_per_side)
    side = np.linspace(offset, 1 - offset, n_per_side)
    return np.stack([np.tile(side[None, :], (n_per_side, 1)), np.tile(side[:, None], (1, n_per_side))], axis=-1).reshape(-1, 2)


def preprocess_shape(oldh, oldw, long_side):
    """ResizeLongestSide.get_preprocess_shape (utils/transforms.py:95-102)."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


@torch.no_grad()
def generate_proposals(prompt_encoder, mask_decoder, image_embedding, original_size, img_size=1024, points_per_side=32,
                       points_per_batch=256, mask_threshold=0.0, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                       stability_score_offset=1.0, box_nms_thresh=0.7):
    """The single-crop path of ``SamAutomaticMaskGenerator._generate_masks`` (crop_n_layers = 0, the configuration the
    ISM uses: model/sam.py:52-75) from the frame's image embedding to the de-duplicated proposals, entirely on the
    device: point grid -> batches of prompts (process_point_batch) -> box NMS on the predicted IoUs.  With one crop
    that spans the frame the crop-edge filter never fires (a box edge near the crop edge is near the image edge too,
    utils/amg.py:78-88) and uncrop_* are identities; the RLE encode / decode round trip of the reference
    (automatic_mask_generator.py:308-310, model/sam.py:146-148) is skipped -- the masks stay binary tensors.
    -> dict(masks bool (K,H,W), boxes (K,4) long XYXY, iou_preds (K,), stability_score (K,), points (K,2) float64)."""
    H, W = original_size
    input_size = preprocess_shape(H, W, img_size)
    grid = build_point_grid(points_per_side) * [[W, H]]                                   # points in the frame (x, y)
    scale = [[input_size[1] / W, input_size[0] / H]]                                      # apply_coords (transforms.py:33-43)
    pts = torch.as_tensor(grid * scale, device=image_embedding.device)                    # float64, as in the reference
    prof = policy.current().amg_profile                                           # per-step milliseconds on stdout
    tq = [time.perf_counter()]

    def tick(name):
        if prof:
            torch.cuda.synchronize()
            tq.append(time.perf_counter())
            print(f"[amg] {name}: {(tq[-1] - tq[-2]) * 1e3:.1f} ms", flush=True)
    parts = []
    for a in range(0, pts.shape[0], points_per_batch):
        r = process_point_batch(prompt_encoder, mask_decoder, image_embedding, pts[a:a + points_per_batch], input_size,
                                original_size, img_size, mask_threshold, pred_iou_thresh, stability_score_thresh,
                                stability_score_offset)
        r["points"] = torch.as_tensor(grid, device=pts.device)[a:a + points_per_batch][r["point_index"]]
        parts.append(r)
    tick("prompt batches")
    cat = {k: torch.cat([p[k] for p in parts]) for k in ("masks", "boxes", "iou_preds", "stability_score", "points")}
    tick(f"concatenate {cat['masks'].shape[0]} candidates")
    keep = ops.nms(cat["boxes"].float().contiguous(), cat["iou_preds"].float(), box_nms_thresh)
    tick("nms")
    out = {k: v[keep] for k, v in cat.items()}
    tick(f"gather {keep.shape[0]} survivors")
    return out


def segmentor_input_size(orig_size, segmentor_width_size):
    """(height, width) the frame is resized to before SAM sees it: CustomSamAutomaticMaskGenerator.preprocess_resize
    (Instance_Segmentation_Model/model/sam.py:75-81; ``segmentor_width_size`` is 640 in configs/model/ISM_sam.yaml).  The
    resize itself is cv2.resize on the host in the reference (bilinear, fixed-point; cv2 is an un-vendored dependency --
    not restated here); for frames that already are that wide -- every 640 x 480 BOP set and the custom demo -- it is the
    identity."""
    return int(segmentor_width_size * orig_size[0] / orig_size[1]), int(segmentor_width_size)


def postprocess_resize(masks, boxes, orig_size, segmentor_width_size):
    """CustomSamAutomaticMaskGenerator.postprocess_resize (model/sam.py:83-100): masks (N,h,w) bool/float found on the
    resized frame -> FLOAT masks (N,H,W) by bilinear interpolation (align_corners=False; the reference keeps the
    fractional edge values, it does not re-threshold), boxes (N,4) -> float, scaled by W / segmentor_width_size and
    clamped to the frame.  Same torch ops as the reference, on whatever device the tensors live."""
    H, W = int(orig_size[0]), int(orig_size[1])
    m = torch.nn.functional.interpolate(masks.unsqueeze(1).float(), size=(H, W), mode="bilinear", align_corners=False)[:, 0]
    b = boxes.float() * (W / segmentor_width_size)
    b[:, [0, 2]] = torch.clamp(b[:, [0, 2]], 0, W - 1)
    b[:, [1, 3]] = torch.clamp(b[:, [1, 3]], 0, H - 1)
    return m, b
