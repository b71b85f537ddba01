"""One frame through every stage in one process, everything on the device (the chain SURVEY.md section 8f points at).

    RGB-D frame -> SAM ViT-H embedding -> 1024 point prompts -> masks / boxes (filters + NMS)           [ISM, segmentor]
               -> masked 224^2 crops -> DINOv2 cls + patch descriptors -> semantic / appearance / geometric scores  [ISM]
               -> top detections -> point clouds + colour crops (pre-processing) -> PEM Net -> R, t per detection [PEM]

The reference runs this as two programs joined by JSON / npz files (``demo.sh``: run_inference_custom.py of the ISM, then
of the PEM) with CPU data loading in between.  ``FramePipeline`` holds the five models and the per-object template data
and passes tensors from stage to stage (``sam6d_amd.ism.handoff.Detections``); each stage is the drop-in module documented
in INTEGRATION.md, so the arithmetic of every stage is the parity-tested one.  What this class adds is only the glue the
reference has in its two ``run_inference_custom.py`` scripts (detector.py:331-430 for the ISM side), reduced to tensor ops:
image resize to the encoder's input (``sam/transforms.py``: Pillow's fixed-point bilinear resampler restated as integer
tensor ops on the device, bit-identical with what ``SamPredictor.set_image`` feeds the model),
dropping tiny detections (``Detections.remove_very_small_detections``, model/utils.py:96-105: box area / frame > 0.05**2 and
mask area / frame > 3e-4 -- the first threshold squared, the second not, as in the reference), keeping the best
detections by final score.
"""
import os
import time
from types import SimpleNamespace

import torch

from . import policy
from .ism.handoff import Detections
from .pem import preprocess as pem_pre
from .sam import amg
from .sam.image_encoder import preprocess as sam_preprocess
from .sam.transforms import ResizeLongestSide


class FramePipeline:
    def __init__(self, sam_encoder, prompt_encoder, mask_decoder, descriptor_model, scorer, pem_net, pem_templates,
                 object_radius, top_k=10, points_per_batch=1024, min_box_size=0.05 ** 2, min_mask_size=3e-4,
                 segmentor=None, nms_per_object_thresh=None, det_score_thresh=None, sync_stages=True):
        """descriptor_model: sam6d_amd.ism.dinov2.CustomDINOv2; scorer: sam6d_amd.ism.scoring.FrameScorer (holds the
        template descriptors); pem_templates: dict(dense_po (O,n,3), dense_fo (O,n,C), model (O,m,3)) of the O objects (O = 1: every detection
        is that object; O > 1: rows are picked by the ISM's predicted object id); object_radius: a number or an (O,) tensor;
        segmentor: keyword overrides of amg.generate_proposals (thresholds).  nms_per_object_thresh: the BOP flow's
        ``apply_nms_per_object_id`` after scoring (detector.py:388-390; 0.25 in configs/model/ISM_sam.yaml; the custom
        demo flow has none).  det_score_thresh: only detections scoring above it go to the PEM
        (run_inference_custom.py:165-171, default 0.2 there); top_k=None keeps every detection, top_k="keys" keeps as many per
        frame as that frame's sample_keys / coarse_rand_u have rows (frames with different instance counts in one group).  sync_stages=False: the
        per-stage timing synchronisations are dropped (``times`` stays empty).  The host still waits on the device where a stage's
        SHAPE depends on data -- the number of proposals surviving the filters / NMS, the crop geometry table of the
        descriptor stage (built on the host from the boxes), the number of detections kept for the PEM, the camera intrinsics
        read as Python floats in pem_pre -- a handful of small copies per frame, so frames issued back to back overlap only
        between those points (measured: 73-76 ms per frame in every mode)."""
        self.enc, self.pe, self.md, self.desc, self.scorer, self.pem = (sam_encoder, prompt_encoder, mask_decoder,
                                                                       descriptor_model, scorer, pem_net)
        self.tpl, self.radius, self.top_k, self.ppb = pem_templates, object_radius, top_k, points_per_batch
        self.min_box, self.min_mask = min_box_size, min_mask_size
        self.seg_kw = segmentor or {}
        self.nms_thresh, self.det_thresh = nms_per_object_thresh, det_score_thresh
        self.times = {}
        self.sync_stages = sync_stages
        self.graph_max = int(policy.current().pem_graph_max)      # instance counts up to this one replay a captured graph
        self._pem_graphs = {}
        # a captured graph bakes the weights' addresses and derived buffers in: the fingerprint in _pem_graph_key sees version bumps and
        # re-allocations, a load_state_dict is caught here (in-place `.data` edits bump no version: call invalidate_graphs())
        if hasattr(self.pem, "register_load_state_dict_post_hook"):
            self.pem.register_load_state_dict_post_hook(lambda module, incompatible: self.invalidate_graphs())

    def score_metres(self, cls, patch, masks, boxes, depth_m, K):
        """The unit boundary between the two halves of the frame: this class takes depth in METRES (what the PEM
        pre-processing needs); the ISM's query translation is Z = depth * depth_scale / 1000 on a millimetre map
        (trimesh_utils.py:87 "depth metric is mm", run_inference_custom.py reads the png as int32 mm), so a map in metres
        goes in with depth_scale 1000."""
        return self.scorer.score(cls, patch, masks, boxes, depth_m, K, depth_scale=1000.0)

    def _tick(self, name, t0):
        if not self.sync_stages:
            return t0
        torch.cuda.synchronize()
        self.times[name] = (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    @torch.no_grad()
    def __call__(self, image_u8, depth, K, sample_keys, coarse_rand_u):
        """image_u8 (H,W,3) uint8 RGB, depth (H,W) f32 metres, K (3,3) float64, all on the device.  sample_keys
        (top_k, H*W) / coarse_rand_u (top_k, 18000): the injected random numbers of the two sampling steps.
        -> (Detections of the frame, dict(pred_R, pred_t, pred_pose_score, kept))."""
        return self.run_group([(image_u8, depth, K, sample_keys, coarse_rand_u)])[0]

    def _embed(self, images):
        """SAM image encoder on a group of equally sized frames in ONE pass -> (F,256,64,64) f32."""
        rs = ResizeLongestSide(self.enc.img_size)
        x = torch.stack([rs.apply_image(im).permute(2, 0, 1) for im in images]).float()
        return self.enc(sam_preprocess(x, self.enc.img_size)).float()

    def _segment(self, emb, image_u8):
        """The segmentor stage of one frame (model/sam.py generate_masks after the encoder) -> dict(masks bool (K,H,W), boxes (K,4)).
        A separate method so that proposals of another source can be joined here (the pixels-to-pose golden adds ten depth
        windows to SAM's proposals: tests/test_gpu_zz_pipeline_e2e.py)."""
        H, W = image_u8.shape[:2]
        return amg.generate_proposals(self.pe, self.md, emb, (H, W), self.enc.img_size, points_per_batch=self.ppb, **self.seg_kw)

    def _propose(self, emb, image_u8):
        """proposals of one frame that survive the size filters and whose crop exists -> (masks, boxes).  emb: (1,256,64,64)."""
        H, W = image_u8.shape[:2]
        prop = self._segment(emb, image_u8)
        area = prop["masks"].flatten(1).sum(1).float() / (H * W)
        b = prop["boxes"].float()
        box_area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) / (H * W)
        keep = (box_area > self.min_box) & (area > self.min_mask)
        # proposals whose crop the reference's CropResizePad cannot produce (it raises on most exactly-square crops and on
        # slivers that vanish in the resize; sam6d_amd/ism/dinov2.py crop_params) are dropped here instead of aborting the frame
        from .ism.dinov2 import crop_valid
        keep &= torch.from_numpy(crop_valid(prop["boxes"].cpu().numpy(), self.desc.proposal_size)).to(keep.device)
        return prop["masks"][keep], prop["boxes"][keep]

    def _score(self, cls, patch, masks, boxes, depth, K):
        """descriptors -> scores -> the frame's Detections, best first."""
        sc = self.score_metres(cls, patch, masks.float(), boxes.float(), depth, K)
        order = torch.argsort(sc["final"], descending=True)
        sel = sc["sel"][order]
        det = Detections(0, 0, masks[sel], boxes[sel], sc["final"][order], sc["pred_obj"][order])
        if self.nms_thresh is not None:
            det.apply_nms_per_object_id(self.nms_thresh)
            det.filter(torch.argsort(det.scores, descending=True, stable=True))       # back to best-first
        if self.det_thresh is not None:
            det.filter(det.scores > self.det_thresh)
        if isinstance(self.top_k, int):
            det.filter(slice(0, self.top_k))
        return det

    def _detect(self, emb, image_u8, depth, K):
        """proposals -> descriptors -> scores -> the frame's Detections, best first (None-free: an empty Detections when nothing
        survives).  emb: this frame's (1,256,64,64) embedding."""
        t0 = time.perf_counter()
        masks, boxes = self._propose(emb, image_u8)
        t0 = self._tick("proposals", t0)
        if masks.shape[0] == 0:
            return Detections(0, 0, masks, boxes, boxes.new_zeros(0), boxes.new_zeros(0))
        cls, patch = self.desc(image_u8, SimpleNamespace(masks=masks.float(), boxes=boxes))      # device frame: no host round trip
        t0 = self._tick("descriptors", t0)
        det = self._score(cls, patch, masks, boxes, depth, K)
        self._tick("scoring", t0)
        return det

    def _detect_group(self, emb, frames):
        """``_detect`` for a group of frames with the descriptor ViT batched ACROSS the frames (CustomDINOv2.frame_batcher: batches
        sized for the GEMM tile grid instead of one ragged batch per frame, launched as soon as they fill so that the device has
        descriptor work queued while the host prepares the next frame's proposals).  Same Detections as per-frame calls.
        Measured (tools/probes/desc_group_ab*.py): the ViT alone runs 21.9 instead of 25.6 ms per 128 crops in batches of 255, but a
        group gains only 0.7 ms per frame (61.4 vs 62.1 ms): frame by frame the ViT starts on a cool socket after the mask decoder
        (23.6 ms), four full batches back to back run at the power limit.  S6D_DESC_GROUP=0: frame by frame (A/B runs)."""
        if len(frames) == 1 or not (hasattr(self.desc, "frame_batcher") and policy.current().desc_group == "1"):
            return [self._detect(emb[i:i + 1], f[0], f[1], f[2]) for i, f in enumerate(frames)]
        t0 = time.perf_counter()
        fb, props = self.desc.frame_batcher(), []
        for i, f in enumerate(frames):
            masks, boxes = self._propose(emb[i:i + 1], f[0])
            props.append((masks, boxes))
            fb.add(f[0], SimpleNamespace(masks=masks.float(), boxes=boxes))
        feats = fb.finish()
        t0 = self._tick("proposals+descriptors", t0)
        dets = []
        for f, (masks, boxes), ft in zip(frames, props, feats):
            if ft is None:
                dets.append(Detections(0, 0, masks, boxes, boxes.new_zeros(0), boxes.new_zeros(0)))
            else:
                dets.append(self._score(ft[0], ft[1], masks, boxes, f[1], f[2]))
        self._tick("scoring", t0)
        return dets

    def _pem_forward(self, ep):
        """Net.forward for the group's M instances.  With few instances the point transformer is LAUNCH-bound (measured: 18.6 ms for
        10 instances against 37 ms for 32, ~1500 launches either way), so for M <= graph_max_instances the forward is captured once
        per instance count as a hipGraph over static input buffers and replayed (no host round trip sits inside Net.forward: the
        coarse uniforms are an input).  S6D_PEM_GRAPH=0 turns it off; counts are padded up to a multiple of 2 by repeating the last
        instance so that a few graphs serve every frame."""
        M = ep["pts"].shape[0]
        dev = ep["pts"].device
        if not (dev.type == "cuda" and M <= self.graph_max and policy.current().pem_graph == "1"):
            return self.pem(ep)
        Mp = (M + 1) // 2 * 2
        keys = ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo", "coarse_rand_u")
        # A captured graph bakes in everything live at capture time: the shapes and dtypes of the inputs, the extractor
        # dtype switch, and the weight-derived buffers the forward caches (hi / lo splits, half copies).  The key carries all of
        # them -- the weights through the sum of their tensor versions and their storage addresses (load_state_dict, in-place
        # updates, .to() all change one of the two) -- so a replay never computes with stale buffers; a stale entry is dropped.
        key = self._pem_graph_key(Mp, ep, keys)
        g = self._pem_graphs.get(Mp)
        if g is not None and g[3] != key:                       # (graph, static inputs, outputs, key, overflow flag)
            del self._pem_graphs[Mp]
            g = None
        if g is None:
            static = {k: torch.empty((Mp,) + tuple(ep[k].shape[1:]), dtype=ep[k].dtype, device=dev) for k in keys}

            def fill():
                for k in keys:
                    static[k][:M].copy_(ep[k])
                    if Mp > M:
                        static[k][M:].copy_(ep[k][M - 1:M].expand(Mp - M, *ep[k].shape[1:]))
            fill()
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(2):                                          # warm every cache the forward fills lazily
                    self.pem(dict(static))
            torch.cuda.current_stream(dev).wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.pem(dict(static))
            g = (graph, static, {k: out[k] for k in ("pred_R", "pred_t", "pred_pose_score")}, key, out.get("f16_overflow"))
            self._pem_graphs[Mp] = g
        graph, static, outs, _, overflow = g
        for k in keys:
            static[k][:M].copy_(ep[k])
            if Mp > M:
                static[k][M:].copy_(ep[k][M - 1:M].expand(Mp - M, *ep[k].shape[1:]))
        graph.replay()
        if overflow is not None and policy.current().pem_f16_guard != "0" and bool(overflow[:M].any()):
            # the IEEE-half extractor overflowed for an instance (Net._f16_range_guard cannot read its flag inside a capture): the
            # eager forward re-runs the flagged instances with the fp32 extractor and warns
            return self.pem(ep)
        return {k: v[:M].clone() for k, v in outs.items()}

    def _pem_graph_key(self, Mp, ep, keys):
        ver, ptr = 0, 0
        for t in list(self.pem.parameters()) + list(self.pem.buffers()):
            ver += t._version
            ptr ^= t.data_ptr()
        return (Mp, tuple((tuple(ep[k].shape[1:]), ep[k].dtype) for k in keys), policy.version(), ver, ptr)

    def invalidate_graphs(self):
        """Drop every captured PEM graph (they are re-captured on the next call)."""
        self._pem_graphs.clear()

    @torch.no_grad()
    def run_group(self, frames):
        """A group of frames through the chain with the batch where the models want it: ONE SAM encoder pass over the group's frames,
        proposals and scores frame by frame (1024 prompts are a full batch already), the descriptor ViT over the group's crops in
        batches that fill the GEMM tile grid (a frame's 128 x 257 token rows are 128.5 row tiles: ragged), the PEM's
        pre-processing frame by frame and ONE PEM pass over the instances of the whole group (10 instances per frame leave
        the point transformer launch-bound; 32 fill it).  frames: list of (image_u8, depth, K, sample_keys, coarse_rand_u) as in
        ``__call__``, all of one size.  -> list of (Detections, poses-or-None) per frame, the same values frame-by-frame calls
        give (a row of a batch does not depend on its neighbours in any kernel of this library; the library's fp32 GEMMs may
        pick another tile shape for another batch size: last-bit differences in the PEM outputs)."""
        for (_, _, _, keys, ru) in frames:
            if isinstance(self.top_k, int) and (keys.shape[0] < self.top_k or ru.shape[0] < self.top_k):
                raise ValueError(f"sample_keys / coarse_rand_u need one row per detection handed to the PEM (top_k={self.top_k}); "
                                 f"got {keys.shape[0]} / {ru.shape[0]}")
        t0 = time.perf_counter()
        emb = self._embed([f[0] for f in frames])
        t0 = self._tick("sam_encoder", t0)
        dets = self._detect_group(emb, frames)
        if self.top_k == "keys":                                         # per-frame instance budget = rows of injected randoms
            for det, f in zip(dets, frames):
                det.filter(slice(0, min(f[3].shape[0], f[4].shape[0])))
        # ---- PEM: pre-processing per frame, one batch for the group ------------------------------------------------------------
        t0 = time.perf_counter()
        multi = self.tpl["model"].shape[0] > 1                          # per-object template data, indexed by predicted object
        obs_l, oid_l, ru_l, count = [], [], [], []
        for det, (image_u8, depth, K, sample_keys, coarse_rand_u) in zip(dets, frames):
            n = 0
            if len(det):
                radius = self.radius
                if torch.is_tensor(radius) and radius.numel() > 1:
                    radius = radius.to(det.object_ids.device)[det.object_ids.long()]
                if len(det) > sample_keys.shape[0] or len(det) > coarse_rand_u.shape[0]:
                    raise ValueError(f"{len(det)} detections go to the PEM but sample_keys / coarse_rand_u have "
                                     f"{sample_keys.shape[0]} / {coarse_rand_u.shape[0]} rows (top_k=None keeps every detection)")
                obs = pem_pre.observed_inputs(image_u8, depth, K, det.masks, radius, sample_keys[: det.masks.shape[0]])
                n = obs["pts"].shape[0]
                if n:
                    obs_l.append(obs)
                    oid_l.append(det.object_ids[obs["kept"]].long())
                    ru_l.append(coarse_rand_u[:n])
            count.append(n)
        t0 = self._tick("pem_preprocessing", t0)
        M = sum(count)
        if M == 0:
            return [(det, None) for det in dets]
        oid = torch.cat(oid_l)

        def tpl(name):                                                  # test_bop.py:145-147 picks dense_po[obj] / dense_fo[obj]
            t = self.tpl[name]
            return (t[oid] if multi else t.expand(M, -1, -1)).contiguous()
        cat = (lambda k: obs_l[0][k]) if len(obs_l) == 1 else (lambda k: torch.cat([o[k] for o in obs_l]))
        ep = dict(pts=cat("pts"), rgb=cat("rgb"), rgb_choose=cat("rgb_choose"), model=tpl("model"), dense_po=tpl("dense_po"),
                  dense_fo=tpl("dense_fo"), coarse_rand_u=ru_l[0] if len(ru_l) == 1 else torch.cat(ru_l))
        out = self._pem_forward(ep)
        self._tick("pem", t0)
        res, at, k = [], 0, 0
        for det, n in zip(dets, count):
            if n == 0:
                res.append((det, None))
                continue
            res.append((det, dict(pred_R=out["pred_R"][at:at + n], pred_t=out["pred_t"][at:at + n],
                                  pred_pose_score=out["pred_pose_score"][at:at + n], kept=obs_l[k]["kept"])))
            at += n
            k += 1
        return res


def frame_results(det, poses, dataset_name, time_s=0.0):
    """The three result files of the reference for one frame, from what ``FramePipeline.__call__`` returned: the ISM JSON
    records (detection_ism.json / result_<ds>.json, sam6d_amd.ism.handoff), and for the detections the PEM kept
    (``poses['kept']`` indexes ``det``) the BOP csv lines and the detection_pem.json records (sam6d_amd.pem.results).
    -> dict(ism_records, csv_lines, pem_records); the last two are empty when ``poses`` is None."""
    from .ism.handoff import detection_records
    from .pem import results

    ism = detection_records(det, dataset_name) if det.masks.shape[0] else []
    if poses is None:
        return dict(ism_records=ism, csv_lines=[], pem_records=[])
    kept = poses["kept"].cpu().tolist()
    s = results.combined_scores(poses["pred_pose_score"], det.scores[poses["kept"]])
    sub = [ism[i] for i in kept]
    csv = results.bop_csv_lines(det.scene_id, det.image_id, [r["category_id"] for r in sub], s, poses["pred_R"],
                                poses["pred_t"], time_s)
    return dict(ism_records=ism, csv_lines=csv, pem_records=results.detection_pem_records(sub, s, poses["pred_R"], poses["pred_t"]))
