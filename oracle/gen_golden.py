"""Generate tests/golden/*.npz by running the REFERENCE's own modules (test infrastructure).

Run in the build container only (needs /root/reference, read-only):

    python -m oracle.gen_golden pem      # Pose_Estimation_Model Net + known-answer case
    python -m oracle.gen_golden sam      # SAM ImageEncoderViT (mini config + ViT-H)
    python -m oracle.gen_golden ism      # ISM scoring chain
    python -m oracle.gen_golden dinov2   # DINOv2 crops + descriptors (next row 8f-1)
    python -m oracle.gen_golden sam_decoder   # SAM prompt encoder + mask decoder (next row 8f-2)

PEM and ISM must run in separate processes (both trees own top-level names).  The
reference modules are imported unmodified through oracle/refharness.py; weights and inputs
come from sam6d_amd.utils.{seeded,synth}, so tests regenerate identical tensors without
this script, /root/reference or any checkpoint.  Large tensors are stored as a strided
sample plus sums (fixtures stay small); small ones are stored whole.
"""
import os
import sys
import types

import numpy as np
import torch

from sam6d_amd.utils import seeded, synth

from . import refharness as rh

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

PEM_CASE = dict(B=2, weight_seed=1, input_seed=1, rand_seed=7)
SAM_MINI_CASE = dict(weight_seed=3, input_seed=5)
SAM_H_CASE = dict(weight_seed=3, input_seed=5)
ISM_CASE = dict(P=64, O=3, T=42, seed=11)
ISM_CASE_BENCH = dict(P=128, O=1, T=42, seed=11)          # the shape bench.py scores per frame (BASELINE configs[1])
ISM_CASE_YCBV = dict(P=128, O=21, T=42, seed=12)          # BASELINE configs[2]: YCB-V, 21 objects
ISM_CASE_TLESS = dict(P=256, O=30, T=42, seed=5)          # BASELINE configs[3]: T-LESS, 30 objects, many-instance frames
SAMDEC_CASE = dict(weight_seed=2, input_seed=9, n_mini=9, n_full=4, mini_input_size=(96, 128), mini_orig=(60, 80),
                   post_B=3, post_seed=6, post_input_size=(768, 1024), post_orig=(480, 640))
DINO_CASE = dict(P=8, input_seed=4, weight_seed=6, mini_target=56, n_full=2)


def digest(t, stride=97):
    """(sum, abs-sum, strided sample) of a big tensor -- the fixture form of a large output."""
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item()]), t[::stride].float().numpy()


def gen_pem():
    ns = rh.pem()
    cfg = rh.pem_cfg()
    net = ns.pose_estimation_model.Net(cfg.model).eval()
    seeded.load_seeded(net, PEM_CASE["weight_seed"])
    B = PEM_CASE["B"]
    inp = synth.pem_inputs(B, seed=PEM_CASE["input_seed"])
    ep = {k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}
    rec = {}
    with torch.no_grad():
        # --- full Net.forward (ViT features are from random weights: pose is arbitrary but fixed)
        torch.manual_seed(PEM_CASE["rand_seed"])  # the reference draws torch.rand(B,18000) from this
        out = net(dict(ep))
        for k in ("init_R", "init_t", "pred_R", "pred_t", "pred_pose_score"):
            rec["net_" + k] = out[k].numpy()
        # --- stages, by calling the reference sub-modules in Net.forward's order
        dense_pm, dense_fm, dense_po, dense_fo, radius = net.feature_extraction(dict(ep))
        rec["fe_radius"] = radius.numpy()
        rec["fe_dense_fm_sum"], rec["fe_dense_fm_smp"] = digest(dense_fm)
        # --- known-answer case: observed features = template features + noise (no ViT)
        dense_fm = inp["dense_fm_kat"]
        bg = torch.ones(B, 1, 3) * 100
        sp_m, sf_m, idx_m = ns.model_utils.sample_pts_feats(dense_pm, dense_fm, 196, return_index=True)
        geo_m = net.geo_embedding(torch.cat([bg, sp_m], 1))
        sp_o, sf_o, idx_o = ns.model_utils.sample_pts_feats(dense_po, dense_fo, 196, return_index=True)
        geo_o = net.geo_embedding(torch.cat([bg, sp_o], 1))
        rec["kat_fps_idx_m"], rec["kat_fps_idx_o"] = idx_m.numpy(), idx_o.numpy()
        rec["kat_geo_m_sum"], rec["kat_geo_m_smp"] = digest(geo_m, 9973)
        e = {"model": ep["model"]}
        torch.manual_seed(PEM_CASE["rand_seed"])
        e = net.coarse_point_matching(sp_m, sf_m, geo_m, sp_o, sf_o, geo_o, radius, e)
        e = net.fine_point_matching(dense_pm, dense_fm, geo_m, idx_m, dense_po, dense_fo, geo_o, idx_o, radius, e)
        for k in ("init_R", "init_t", "pred_R", "pred_t", "pred_pose_score"):
            rec["kat_" + k] = e[k].numpy()
        rec["kat_gt_R"], rec["kat_gt_t"] = inp["gt_R"].numpy(), inp["gt_t"].numpy()
        # --- PositionalEncoding alone (ball query + grouping + SharedMLP + max-pool)
        pe = net.fine_point_matching.PE(dense_po)
        rec["pe_sum"], rec["pe_smp"] = digest(pe, 997)
    rec["state_keys"] = np.array(sorted(net.state_dict().keys()))
    rec["state_shapes"] = np.array([str(tuple(net.state_dict()[k].shape)) for k in sorted(net.state_dict().keys())])
    rec["case"] = np.array(str(PEM_CASE))
    np.savez_compressed(os.path.join(OUT, "pem_b2.npz"), **rec)
    print("pem_b2.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in rec.items() if not k.startswith("state")})
    dR = np.linalg.norm(rec["kat_pred_R"] - rec["kat_gt_R"], axis=(1, 2))
    print("KAT |dR|_F", dR, "score", rec["kat_pred_pose_score"], "t", rec["kat_pred_t"])


PEM_WC_CASE = dict(B=2, weight_seed=1, input_seed=91, rand_seed=92)


def gen_pem_wc():
    """Net.forward on a WELL-CONDITIONED frame (tests/golden/pem_wc.npz): the template features are the reference feature
    extractor's own output for the observed pixels (dense_fo = dense_fm), so observed point i matches template point i the way
    a rendered template matches the object it shows.  (The pem_b2 `net_*` case feeds template features that are unrelated
    to the random-weight ViT's output: its pose is the arg-max over near-tied hypotheses and flips under a 1e-3 relative
    perturbation of the features -- see tests/test_host_pem.py::test_conditioning_of_the_two_net_forward_cases.)  Stored:
    the reference Net's pose; the test recomputes dense_fo with the oracle's feature extractor, which this script checks to be
    bit-identical with the reference's on this host."""
    from . import pem as opem
    ns = rh.pem()
    cfg = rh.pem_cfg()
    net = ns.pose_estimation_model.Net(cfg.model).eval()
    seeded.load_seeded(net, PEM_WC_CASE["weight_seed"])
    B = PEM_WC_CASE["B"]
    inp = synth.pem_inputs(B, seed=PEM_WC_CASE["input_seed"])
    ep = {k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}
    rec = {}
    with torch.no_grad():
        _, dense_fm, _, _, _ = net.feature_extraction(dict(ep))
        W = {k: v.clone() for k, v in net.state_dict().items()}
        assert torch.equal(opem.feature_extraction(W, ep)[1], dense_fm), "oracle feature extractor != reference on this host"
        ep["dense_fo"] = dense_fm.clone()
        torch.manual_seed(PEM_WC_CASE["rand_seed"])
        out = net(dict(ep))
        for k in ("init_R", "init_t", "pred_R", "pred_t", "pred_pose_score"):
            rec["net_" + k] = out[k].numpy()
    rec["fo_sum"], rec["fo_smp"] = digest(dense_fm, 4099)
    rec["gt_R"], rec["gt_t"] = inp["gt_R"].numpy(), inp["gt_t"].numpy()
    rec["case"] = np.array(str(PEM_WC_CASE))
    np.savez_compressed(os.path.join(OUT, "pem_wc.npz"), **rec)
    print("pem_wc.npz |dR|_F vs truth", np.linalg.norm(rec["net_pred_R"] - rec["gt_R"], axis=(1, 2)),
          "|dt|", np.abs(rec["net_pred_t"] - rec["gt_t"]).max(), "score", rec["net_pred_pose_score"])


def gen_pem_b32():
    """The matching path (coarse + fine point matching, pose solvers) of the reference Net at the BENCHED batch (B = 32): the
    reference sub-modules in Net.forward's order on the known-answer inputs, outputs only (tests/golden/pem_b32.npz)."""
    ns = rh.pem()
    cfg = rh.pem_cfg()
    net = ns.pose_estimation_model.Net(cfg.model).eval()
    seeded.load_seeded(net, PEM_CASE["weight_seed"])
    case = dict(B=32, input_seed=41, rand_seed=42, weight_seed=PEM_CASE["weight_seed"])
    B = case["B"]
    inp = synth.pem_inputs(B, seed=case["input_seed"], with_rgb=False)
    rec = {}
    with torch.no_grad():
        radius = torch.norm(inp["dense_po"], dim=2).max(1)[0]
        s = radius.reshape(-1, 1, 1) + 1e-6
        dense_pm, dense_po, dense_fm, dense_fo = inp["pts"] / s, inp["dense_po"] / s, inp["dense_fm_kat"], inp["dense_fo"]
        bg = torch.ones(B, 1, 3) * 100
        sp_m, sf_m, idx_m = ns.model_utils.sample_pts_feats(dense_pm, dense_fm, 196, return_index=True)
        geo_m = net.geo_embedding(torch.cat([bg, sp_m], 1))
        sp_o, sf_o, idx_o = ns.model_utils.sample_pts_feats(dense_po, dense_fo, 196, return_index=True)
        geo_o = net.geo_embedding(torch.cat([bg, sp_o], 1))
        e = {"model": inp["model"]}
        torch.manual_seed(case["rand_seed"])  # the reference draws torch.rand(B, 18000) from this (= synth.coarse_uniforms)
        e = net.coarse_point_matching(sp_m, sf_m, geo_m, sp_o, sf_o, geo_o, radius, e)
        e = net.fine_point_matching(dense_pm, dense_fm, geo_m, idx_m, dense_po, dense_fo, geo_o, idx_o, radius, e)
        for k in ("init_R", "init_t", "pred_R", "pred_t", "pred_pose_score"):
            rec["kat_" + k] = e[k].numpy()
        rec["kat_gt_R"], rec["kat_gt_t"] = inp["gt_R"].numpy(), inp["gt_t"].numpy()
    rec["case"] = np.array(str(case))
    np.savez_compressed(os.path.join(OUT, "pem_b32.npz"), **rec)
    dR = np.linalg.norm(rec["kat_pred_R"] - rec["kat_gt_R"], axis=(1, 2))
    print("pem_b32.npz KAT |dR|_F max", dR.max(), "score min", rec["kat_pred_pose_score"].min())


def _sam_ref(cfg, weight_seed):
    enc = rh.sam_encoder()
    from functools import partial
    m = enc.ImageEncoderViT(
        depth=cfg["depth"], embed_dim=cfg["dim"], img_size=cfg["img_size"], mlp_ratio=4,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=cfg["heads"], patch_size=16,
        qkv_bias=True, use_rel_pos=True, global_attn_indexes=cfg["global_idx"], window_size=cfg["window"],
        out_chans=cfg["out_chans"]).eval()
    seeded.load_seeded(m, weight_seed)
    return m


def gen_sam():
    from . import sam as osam
    rec = {}
    with torch.no_grad():
        m = _sam_ref(osam.MINI, SAM_MINI_CASE["weight_seed"])
        x = synth.sam_input(1, SAM_MINI_CASE["input_seed"], osam.MINI["img_size"])
        rec["mini_out"] = m(x).numpy()
        rec["mini_keys"] = np.array(sorted(m.state_dict().keys()))
        m = _sam_ref(osam.VIT_H, SAM_H_CASE["weight_seed"])
        x = synth.sam_input(1, SAM_H_CASE["input_seed"], 1024)
        y = m(x)
        rec["h_sum"], rec["h_smp"] = digest(y, 251)
        rec["h_keys"] = np.array(sorted(m.state_dict().keys()))
        rec["h_shapes"] = np.array([str(tuple(m.state_dict()[k].shape)) for k in sorted(m.state_dict().keys())])
        # token map after 2 blocks (one windowed-with-padding pair) for a per-block check
        t = m.patch_embed(x) + m.pos_embed
        for blk in m.blocks[:2]:
            t = blk(t)
        rec["h_blk2_sum"], rec["h_blk2_smp"] = digest(t, 1009)
    np.savez_compressed(os.path.join(OUT, "sam_enc.npz"), **rec)
    print("sam_enc.npz", {k: v.shape for k, v in rec.items()})


def gen_ism():
    ns = rh.ism()
    _gen_ism(ns, ISM_CASE, "ism_scoring.npz")
    _gen_ism(ns, ISM_CASE_BENCH, "ism_scoring_p128.npz")
    _gen_ism(ns, ISM_CASE_YCBV, "ism_scoring_ycbv.npz", compact=True)
    _gen_ism(ns, ISM_CASE_TLESS, "ism_scoring_tless.npz", compact=True)


def _gen_ism(ns, c, fname, compact=False):
    inp = synth.ism_inputs(P=c["P"], O=c["O"], T=c["T"], seed=c["seed"])
    Det = ns.detector.Instance_Segmentation_Model
    fake = types.SimpleNamespace()
    fake.ref_data = dict(descriptors=inp["ref_cls"], appe_descriptors=inp["ref_patch"], poses=inp["poses"],
                         pointcloud=inp["pointcloud"])
    fake.matching_config = types.SimpleNamespace(metric=ns.loss.PairwiseSimilarity(), aggregation_function="avg_5",
                                                 confidence_thresh=0.2)
    for name in ("best_template_pose", "compute_semantic_score", "compute_appearance_score",
                 "compute_geometric_score", "project_template_to_image", "Calculate_the_query_translation"):
        setattr(fake, name, types.MethodType(getattr(Det, name), fake))
    rec = {}
    with torch.no_grad():
        rec["pairwise"] = fake.matching_config.metric(inp["qry_cls"], inp["ref_cls"]).numpy()
        sel, pobj, sem, bt = fake.compute_semantic_score(inp["qry_cls"])
        qp = inp["qry_patch"][sel]
        appe, ref = fake.compute_appearance_score(bt, pobj, qp)
        batch = dict(depth=[inp["depth"]], cam_intrinsic=[inp["K"]], depth_scale=1.0)
        uv = fake.project_template_to_image(bt, pobj, batch, inp["masks"][sel].clone())
        dets = types.SimpleNamespace(boxes=inp["boxes"][sel])
        geo, vr = fake.compute_geometric_score(uv, dets, qp, ref, visible_thred=0.5)
        final = (sem + appe + geo * vr) / (1 + 1 + vr)
        # second geometry case: boxes that all overlap the projections -> non-degenerate IoU (quirk Q3 off)
        xyxy = torch.cat((uv.min(1).values, uv.max(1).values), -1).float()
        boxes2 = xyxy + torch.tensor([-3.0, -2.0, 4.0, 5.0])
        iou2 = ns.bbox_utils.compute_iou(xyxy, boxes2)
        rec.update(sel=sel.numpy(), pred_obj=pobj.numpy(), semantic=sem.numpy(), best_template=bt.numpy(),
                   appearance=appe.numpy(), image_uv=uv.numpy(), visible_ratio=vr.numpy(),
                   iou=np.asarray(geo if not torch.is_tensor(geo) else geo.numpy(), dtype=np.float32),
                   final=final.numpy(), iou2=iou2.numpy(), boxes2=boxes2.numpy(),
                   translation=fake.Calculate_the_query_translation(inp["masks"][sel].clone(), inp["depth"], inp["K"], 1.0).numpy())
    if compact:
        # the large configurations: pixels fit int16 (checked), the (P, O*T) cosine matrix is kept as a strided sample + sums
        uv = rec["image_uv"]
        assert np.abs(uv).max() < 2 ** 15
        rec["image_uv"] = uv.astype(np.int16)
        rec["pairwise_sums"], rec["pairwise_sample"] = digest(torch.from_numpy(rec.pop("pairwise")), 13)
    rec["case"] = np.array(str(c))
    np.savez_compressed(os.path.join(OUT, fname), **rec)
    print(fname, {k: (v.shape if hasattr(v, "shape") else v) for k, v in rec.items()})
    print("selected", len(rec["sel"]), "iou", rec["iou"], "vr", rec["visible_ratio"][:5], "final", rec["final"][:5])


def gen_handoff():
    """Reference mask_to_rle / Detections.save_to_file / convert_npz_to_json (model/utils.py) and the PEM side's
    rle_to_binary_mask (Pose_Estimation_Model/utils/data_utils.py), run unmodified on synthetic detections."""
    import importlib
    import json
    import tempfile
    rh.ism()
    mu = importlib.import_module("model.utils")
    g = torch.Generator().manual_seed(12)
    H, W, N = 37, 53, 6
    masks = (torch.rand(N, H, W, generator=g) > 0.55)
    masks[0] = False                                           # empty mask
    masks[1] = True                                            # full mask (first pixel set: leading zero run of 0)
    masks[2, :, :7] = True
    boxes = torch.tensor([[0, 0, 0, 0], [0, 0, W - 1, H - 1], [0, 0, 6, H - 1], [3, 4, 30, 20], [10, 2, 50, 36], [1, 1, 2, 2]])
    scores = torch.rand(N, generator=g)
    obj = torch.tensor([0, 1, 2, 3, 7, 5])
    rec = {}
    rles = [mu.mask_to_rle(m.numpy().astype(np.uint8)) for m in masks]
    for ds in ("lmo", "ycbv"):
        det = mu.Detections({"masks": masks.clone(), "boxes": boxes.clone(), "scores": scores.clone(), "object_ids": obj.clone()})
        det.to_numpy()
        with tempfile.TemporaryDirectory() as d:
            det.save_to_file(2, 17, 0.25, d + "/f", ds)
            out = mu.convert_npz_to_json(0, [d + "/f.npz"])
        rec[ds + "_json"] = np.array(json.dumps(out))
    rec["rle_json"] = np.array(json.dumps(rles))
    np.savez_compressed(os.path.join(OUT, "handoff.npz"), **rec)
    print("handoff.npz", {k: len(str(v)) for k, v in rec.items()})


def _ref_statements(path, first, last):
    """The statements of a reference script between the first line containing `first` and the first later line containing
    `last` (inclusive), dedented -- for code that lives inline in an entry script and cannot be imported."""
    import textwrap
    lines = open(path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if first in l)
    b = next(i for i in range(a, len(lines)) if last in lines[i])
    return textwrap.dedent("\n".join(lines[a:b + 1]))


def gen_pem_results():
    """The reference's pose writers, executed from its own files on synthetic model outputs: the BOP csv lines of
    Pose_Estimation_Model/test_bop.py ("# write results" .. "lines.append(line)", with the float32 scaling statements just
    above them) and detection_pem.json of run_inference_custom.py ("if 'pred_pose_score' in out.keys()" .. json.dump)."""
    import json
    import tempfile
    pem = os.path.join(rh.REF_ROOT, "SAM-6D", "Pose_Estimation_Model")
    g = torch.Generator().manual_seed(77)
    n = 5
    R = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))[0]
    t = torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 0.8])
    pose_score = torch.rand(n, generator=g)
    det_score = torch.rand(n, generator=g)
    det_score[0] = 0.1
    pose_score[0] = 1.0
    obj = torch.tensor([1, 5, 6, 8, 12])
    rec = dict(R=R.numpy(), t=t.numpy(), pose_score=pose_score.numpy(), det_score=det_score.numpy(), obj=obj.numpy())
    # ---- test_bop.py ----------------------------------------------------------------------------------------------
    src = _ref_statements(os.path.join(pem, "test_bop.py"), "pred_Rs = torch.cat(pred_Rs", "image_time = time.time() - end")
    src2 = _ref_statements(os.path.join(pem, "test_bop.py"), "# write results", "lines.append(line)")
    import time
    ns = dict(torch=torch, time=time, end=time.time(), pred_Rs=[R[:2], R[2:]], pred_Ts=[t[:2], t[2:]], pred_scores=[pose_score[:2], pose_score[2:]],
              data=dict(score=det_score.reshape(1, n, 1), scene_id=torch.tensor([48]), img_id=torch.tensor([1003]), seg_time=torch.tensor([0.0]),
                        obj_id=obj.reshape(1, n)), n_instance=n, lines=[])
    exec(src, ns)
    ns["image_time"] = 0.375                                                           # the wall clock is not reproducible
    exec(src2, ns)
    rec["csv"] = np.array("".join(ns["lines"]))
    # ---- run_inference_custom.py --------------------------------------------------------------------------------------
    src = _ref_statements(os.path.join(pem, "run_inference_custom.py"), "if 'pred_pose_score' in out.keys():", "json.dump(detections, f)")
    dets = [dict(scene_id=0, image_id=0, category_id=int(o), bbox=[1, 2, 30 + i, 40], score=float(det_score[i]), time=0.0,
                 segmentation=dict(counts=[3, 4, 5], size=[3, 4])) for i, o in enumerate(obj)]
    rec["dets_json"] = np.array(json.dumps(dets))
    with tempfile.TemporaryDirectory() as d:
        ns = dict(os=os, json=json, out=dict(pred_pose_score=pose_score, score=det_score, pred_R=R, pred_t=t),
                  cfg=types.SimpleNamespace(output_dir=d), detections=[dict(x) for x in dets])
        exec(src, ns)
        rec["pem_json"] = np.array(open(os.path.join(d, "sam6d_results", "detection_pem.json")).read())
    np.savez_compressed(os.path.join(OUT, "pem_results.npz"), **rec)
    print("pem_results.npz"); print(str(rec["csv"])[:400]); print(str(rec["pem_json"])[:300])


def gen_detections_ops():
    """The reference's Detections list operations (model/utils.py: remove_very_small_detections, apply_nms,
    apply_nms_per_object_id) and CustomSamAutomaticMaskGenerator.postprocess_resize (model/sam.py), run unmodified.
    torchvision is not installed: its two functions the code calls are supplied from their published definitions
    (box_area = (x2-x1)*(y2-y1); nms = oracle/sam_decoder.py nms) -- the NMS boundary stays "parity unpinned"."""
    import importlib
    from . import sam_decoder as od
    rh.ism()
    mu = importlib.import_module("model.utils")
    mu.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    mu.torchvision.ops.nms = od.nms
    g = torch.Generator().manual_seed(21)
    N, H, W = 40, 48, 64
    xy = (torch.rand(N, 2, generator=g) * torch.tensor([W - 20.0, H - 20.0])).floor()
    wh = (2 + torch.rand(N, 2, generator=g) * 18).floor()
    boxes = torch.cat([xy, xy + wh], 1).long()
    boxes[5:9] = boxes[4]                                                   # duplicates: NMS must drop them within an object
    masks = torch.zeros(N, H, W, dtype=torch.bool)
    for i, (x1, y1, x2, y2) in enumerate(boxes.tolist()):
        masks[i, y1:y2, x1:x2] = torch.rand(y2 - y1, x2 - x1, generator=g) > 0.4
    masks[3] = False
    scores = torch.rand(N, generator=g)
    scores[6] = scores[4]                                                   # a score tie between duplicates
    obj = torch.randint(0, 5, (N,), generator=g)
    rec = dict(boxes=boxes.numpy(), masks=np.packbits(masks.numpy()), scores=scores.numpy(), obj=obj.numpy(), shape=np.array([N, H, W]))

    def fresh():
        return mu.Detections({"masks": masks.clone(), "boxes": boxes.clone(), "scores": scores.clone(), "object_ids": obj.clone()})
    d = fresh()
    d.remove_very_small_detections(types.SimpleNamespace(min_box_size=0.05, min_mask_size=3e-2))
    rec["small_scores"] = d.scores.numpy()
    d = fresh()
    d.apply_nms(0.5)
    rec["nms_scores"] = d.scores.numpy()
    d = fresh()
    d.apply_nms_per_object_id(0.25)
    rec["nms_obj_scores"], rec["nms_obj_ids"], rec["nms_obj_boxes"] = d.scores.numpy(), d.object_ids.numpy(), d.boxes.numpy()
    rec["nms_obj_mask_sums"] = d.masks.sum(dim=(1, 2)).numpy()
    # ---- postprocess_resize ----------------------------------------------------------------------------------------
    ms = importlib.import_module("model.sam")
    for tag, orig in (("same", (48, 64)), ("up", (81, 108))):
        det = {"masks": masks[:6].clone(), "boxes": boxes[:6].clone()}
        out = ms.CustomSamAutomaticMaskGenerator.postprocess_resize(types.SimpleNamespace(segmentor_width_size=64), det, orig)
        rec["pp_masks_" + tag], rec["pp_boxes_" + tag] = out["masks"].numpy(), out["boxes"].numpy()
    np.savez_compressed(os.path.join(OUT, "detections_ops.npz"), **rec)
    print("detections_ops.npz", {k: getattr(v, "shape", v) for k, v in rec.items()})


def gen_sam_transforms():
    """The reference's ResizeLongestSide (segment_anything/utils/transforms.py) run unmodified.  torchvision is not
    installed; the two functions apply_image calls are supplied from their documented behaviour on PIL inputs:
    to_pil_image(uint8 HWC ndarray) = PIL.Image.fromarray, resize(pil, (h, w)) = pil.resize((w, h), BILINEAR).  Pillow
    itself is real (installed), so the pixel arithmetic in the golden is Pillow's."""
    from PIL import Image
    ns = rh.sam_decoder()
    tr = ns.transforms
    tr.to_pil_image = lambda a: Image.fromarray(a)
    tr.resize = lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR)
    rng = np.random.default_rng(5)
    rec = {}
    for tag, (H, W), L in (("vga", (60, 80), 128), ("tless", (54, 72), 96), ("itodd", (96, 128), 64), ("tall", (75, 31), 96)):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        img[: H // 2, : W // 2] = (img[: H // 2, : W // 2] // 64) * 64           # flat regions + noise
        t = tr.ResizeLongestSide(L)
        pts = rng.uniform(0, [W, H], (7, 2))
        boxes = np.concatenate([pts[:3], pts[:3] + rng.uniform(1, 20, (3, 2))], 1)
        rec[tag + "_img"], rec[tag + "_L"] = img, np.array(L)
        rec[tag + "_out"] = t.apply_image(img)
        rec[tag + "_pts"], rec[tag + "_pts_out"] = pts, t.apply_coords(pts, (H, W))
        rec[tag + "_boxes"], rec[tag + "_boxes_out"] = boxes, t.apply_boxes(boxes, (H, W))
        rec[tag + "_pts_out_t"] = t.apply_coords_torch(torch.from_numpy(pts), (H, W)).numpy()
        rec[tag + "_boxes_out_t"] = t.apply_boxes_torch(torch.from_numpy(boxes), (H, W)).numpy()
    np.savez_compressed(os.path.join(OUT, "sam_transforms.npz"), **rec)
    print("sam_transforms.npz", {k: getattr(v, "shape", v) for k, v in rec.items() if k.endswith("_out")})


def gen_sam_state_dict():
    """state_dict keys / shapes of the reference's three Sam builds (build_sam.py run unmodified on the meta device)."""
    import json
    mod = rh.sam_builder()
    rec = {}
    for name in ("vit_h", "vit_l", "vit_b"):
        with torch.device("meta"):
            m = mod.sam_model_registry[name]()
        rec[name] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    with open(os.path.join(OUT, "sam_state_dict.json"), "w") as f:
        json.dump(rec, f, separators=(",", ":"))
    print("sam_state_dict.json", {k: len(v) for k, v in rec.items()})


def gen_pem_pre():
    """Reference geometry helpers of the PEM pre-processing (utils/data_utils.py), run unmodified."""
    du = rh.pem_data_utils()
    inp = synth.pem_pre_inputs(P=8, seed=3)
    masks = inp["masks"].numpy()
    depth, K = inp["depth"].numpy(), inp["K"].numpy()
    rec = {"bbox": np.array([du.get_bbox(np.logical_and(m, depth > 0)) for m in masks])}
    cloud = du.get_point_cloud_from_depth(depth, K)                      # float64 under NumPy 2, float32 under 1.x
    rec["cloud_sum"], rec["cloud_smp"] = digest(torch.from_numpy(cloud.astype(np.float32)), 499)
    y1, y2, x1, x2 = rec["bbox"][1]
    rec["cloud_crop_smp"] = du.get_point_cloud_from_depth(depth, K, [y1, y2, x1, x2]).astype(np.float32).reshape(-1, 3)[::61]
    ch = np.arange(0, (y2 - y1) * (x2 - x1), 37)
    rec["rgb_choose"] = du.get_resize_rgb_choose(ch, [y1, y2, x1, x2], 224)
    # the reference's sampling statements (inline in run_inference_custom.py:224-229 and bop_test_dataset.py:140-145),
    # executed from its files with numpy's global RNG seeded: pins how many draws a detection consumes and in which form
    pem = os.path.join(rh.REF_ROOT, "SAM-6D", "Pose_Estimation_Model")
    counts = [40, 700, 512, 513, 5]
    for tag, path, ns_extra in (("custom", os.path.join(pem, "run_inference_custom.py"), dict(cfg=types.SimpleNamespace(n_sample_observed_point=512))),
                                ("bop", os.path.join(pem, "provider", "bop_test_dataset.py"), dict(self=types.SimpleNamespace(n_sample_observed_point=512)))):
        src = _ref_statements(path, "n_sample_observed_point:", "cloud = cloud[choose_idx]")
        np.random.seed(11)
        got = []
        for c in counts:
            ns = dict(np=np, choose=np.arange(c) * 3, cloud=np.arange(c * 3, dtype=np.float32).reshape(c, 3), **ns_extra)
            exec(src, ns)
            got.append(ns["choose"] // 3)
        rec["rng_idx_" + tag] = np.stack(got)
    rec["rng_counts"] = np.array(counts)
    np.savez_compressed(os.path.join(OUT, "pem_pre.npz"), **rec)
    print("pem_pre.npz", {k: v.shape for k, v in rec.items()}, rec["bbox"][:4].tolist())


def gen_example():
    """BASELINE configs[0]: the ONE real frame the reference ships (SAM-6D/Data/Example: rgb.png, depth.png in millimetres,
    camera.json, obj_000005.ply in millimetres; demo.sh:1-17 runs it through both models).  No checkpoint and no ISM result are
    reachable offline, so the plumbing is pinned as far as tensors can be frozen: the frame itself, a deterministic object mask
    (a depth window inside a box around the object at (370, 200)), the reference's own pre-processing helpers
    (utils/data_utils.py get_bbox / get_point_cloud_from_depth / get_resize_rgb_choose, run unmodified) on that mask, model and
    template points taken from the ply's vertices (the reference samples the mesh surface with trimesh's RNG), seeded template
    features and weights.  tests/golden/example_frame.npz holds the inputs (the GPU box has no /root/reference) and these outputs;
    the oracle's pre-processing + Net.forward on them is added by tests/test_oracle_golden.py's twin in gen (`oracle_*` keys)."""
    from PIL import Image

    from . import pem as opem
    from . import pem_pre as opre
    ex = os.path.join(rh.REF_ROOT, "SAM-6D", "Data", "Example")
    rgb = np.array(Image.open(os.path.join(ex, "rgb.png")))[..., :3].astype(np.uint8)
    depth_mm = np.array(Image.open(os.path.join(ex, "depth.png"))).astype(np.uint16)
    import json
    cam = json.load(open(os.path.join(ex, "camera.json")))
    K = np.array(cam["cam_K"], dtype=np.float64).reshape(3, 3)
    # ascii ply: vertex lines after end_header (x y z nx ny nz r g b a), millimetres
    with open(os.path.join(ex, "obj_000005.ply")) as f:
        lines = f.read().split("end_header\n")[1].split("\n")
    nv = 22831
    verts = np.array([[float(v) for v in ln.split()[:3]] for ln in lines[:nv]], dtype=np.float32) / np.float32(1000.0)
    model = np.ascontiguousarray(verts[::22][:1024])                       # (1024,3) m: `model` of the PEM
    dense_po = np.ascontiguousarray(verts[::11][:2048])                    # (2048,3) m: template points
    radius = float(np.max(np.linalg.norm(model, axis=1)))
    depth = depth_mm.astype(np.float32) * np.float32(cam["depth_scale"]) / np.float32(1000.0)        # run_inference_custom.py:203
    mask = np.zeros(depth.shape, bool)
    mask[140:262, 300:442] = (depth_mm[140:262, 300:442] > 900) & (depth_mm[140:262, 300:442] < 1075)
    du = rh.pem_data_utils()
    rec = dict(rgb=rgb, depth_mm=depth_mm, K=K, depth_scale=np.float64(cam["depth_scale"]), model=model, dense_po=dense_po,
               radius=np.float64(radius), mask_box=np.array([140, 262, 300, 442]), mask_window_mm=np.array([900, 1075]))
    m = np.logical_and(mask, depth > 0)
    y1, y2, x1, x2 = du.get_bbox(m)
    rec["ref_bbox"] = np.array([y1, y2, x1, x2])
    cloud = du.get_point_cloud_from_depth(depth, K, [y1, y2, x1, x2]).astype(np.float32).reshape(-1, 3)
    rec["ref_cloud_sum"], rec["ref_cloud_smp"] = digest(torch.from_numpy(cloud), 211)
    ch = m[y1:y2, x1:x2].astype(np.float32).flatten().nonzero()[0]
    rec["ref_n_mask"] = np.int64(len(ch))
    rec["ref_rgb_choose"] = du.get_resize_rgb_choose(ch[::17], [y1, y2, x1, x2], 224)
    # ---- the oracle's full pre-processing and Net.forward on the frozen inputs (seeded weights / template features / uniforms)
    case = dict(weight_seed=PEM_CASE["weight_seed"], key_seed=7, feat_seed=8, rand_seed=9)
    keys = torch.rand(1, depth.size, generator=torch.Generator().manual_seed(case["key_seed"])).numpy()
    obs = opre.preprocess_frame(rgb, depth, K, mask[None], radius, keys=keys)
    dense_fo = torch.randn(1, 2048, 256, generator=torch.Generator().manual_seed(case["feat_seed"]))
    ns = rh.pem()
    net = ns.pose_estimation_model.Net(rh.pem_cfg().model).eval()
    seeded.load_seeded(net, case["weight_seed"])
    W = {k: v.clone() for k, v in net.state_dict().items()}
    ep = dict(pts=torch.from_numpy(obs["pts"]), rgb=torch.from_numpy(obs["rgb"]), rgb_choose=torch.from_numpy(obs["rgb_choose"]),
              model=torch.from_numpy(model)[None], dense_po=torch.from_numpy(dense_po)[None], dense_fo=dense_fo)
    with torch.no_grad():
        out = opem.net_forward(W, ep, synth.coarse_uniforms(1, case["rand_seed"]))
        torch.manual_seed(case["rand_seed"])
        ref = net(dict(ep))                                                 # the reference Net itself on the same tensors
    for k in ("pts", "rgb_choose", "bbox", "kept"):
        rec["oracle_" + k] = obs[k]
    rec["oracle_rgb_sum"], rec["oracle_rgb_smp"] = digest(torch.from_numpy(obs["rgb"]), 4099)
    for k in ("pred_R", "pred_t", "pred_pose_score"):
        rec["oracle_" + k] = out[k].numpy()
        rec["ref_" + k] = ref[k].numpy()
    rec["case"] = np.array(str(case))
    np.savez_compressed(os.path.join(OUT, "example_frame.npz"), **rec)
    print("example_frame.npz bbox", rec["ref_bbox"], "mask px", int(rec["ref_n_mask"]), "radius", radius, "kept", obs["kept"],
          "pred_t", rec["oracle_pred_t"], "|R_oracle - R_ref|", np.abs(rec["oracle_pred_R"] - rec["ref_pred_R"]).max())


FRAME_CASE = dict(P=10, O=1, T=6, C=128, n_patch=64, seed=21, det_score_thresh=0.46, weight_seed=1, rng_seed=31, feat_seed=32,
                  rand_seed=33)


def _frame_inputs():
    """tests/util.frame_inputs: the frame, the proposals and the synthetic descriptors, shared with the tests."""
    from tests import util as tutil
    return tutil.frame_inputs(FRAME_CASE)


def gen_frame_ism(tmp):
    """Stage 1 of the whole-frame golden (ISM process): the reference's statements of run_inference_custom.py:155-199 --
    compute_semantic_score -> filter -> compute_appearance_score -> project_template_to_image -> compute_geometric_score -> final
    score -> Detections.add_attribute / to_numpy / save_to_file -> convert_npz_to_json -- on the frame of _frame_inputs()."""
    import importlib
    import json
    ns = rh.ism()
    mu = importlib.import_module("model.utils")
    inp = _frame_inputs()
    Det = ns.detector.Instance_Segmentation_Model
    fake = types.SimpleNamespace()
    fake.ref_data = dict(descriptors=inp["ref_cls"], appe_descriptors=inp["ref_patch"], poses=inp["poses"], pointcloud=inp["pointcloud"])
    fake.matching_config = types.SimpleNamespace(metric=ns.loss.PairwiseSimilarity(), aggregation_function="avg_5", confidence_thresh=0.2)
    for name in ("best_template_pose", "compute_semantic_score", "compute_appearance_score", "compute_geometric_score",
                 "project_template_to_image", "Calculate_the_query_translation"):
        setattr(fake, name, types.MethodType(getattr(Det, name), fake))
    with torch.no_grad():
        detections = mu.Detections({"masks": inp["masks"].clone(), "boxes": inp["boxes"].clone()})
        sel, pobj, sem, bt = fake.compute_semantic_score(inp["qry_cls"])
        detections.filter(sel)
        qp = inp["qry_patch"][sel, :]
        appe, ref = fake.compute_appearance_score(bt, pobj, qp)
        batch = dict(depth=[inp["depth_mm"]], cam_intrinsic=[inp["K"]], depth_scale=inp["depth_scale"])
        uv = fake.project_template_to_image(bt, pobj, batch, detections.masks)
        geo, vr = fake.compute_geometric_score(uv, detections, qp, ref, visible_thred=0.5)
        final = (sem + appe + geo * vr) / (1 + 1 + vr)
        detections.add_attribute("scores", final)
        detections.add_attribute("object_ids", torch.zeros_like(final))
        detections.to_numpy()
        detections.save_to_file(0, 0, 0, os.path.join(tmp, "detection_ism"), "Custom", return_results=False)
        records = mu.convert_npz_to_json(idx=0, list_npz_paths=[os.path.join(tmp, "detection_ism.npz")])
    json.dump(records, open(os.path.join(tmp, "detection_ism.json"), "w"))
    np.savez(os.path.join(tmp, "ism.npz"), sel=sel.numpy(), final=final.numpy(), semantic=sem.numpy(), appearance=appe.numpy(),
             visible_ratio=vr.numpy(), iou=np.asarray(geo if not torch.is_tensor(geo) else geo.numpy(), dtype=np.float32) * np.ones(len(sel), np.float32),
             image_uv=uv.numpy(), best_template=bt.numpy())
    print("frame/ism: selected", sel.tolist(), "final", final.numpy().round(4).tolist())


def gen_frame_pem(tmp):
    """Stage 2 (PEM process): run_inference_custom.py get_test_data's score threshold (:169-171) on the ISM JSON, masks decoded
    with the reference's rle_to_binary_mask, the per-detection loop (oracle/pem_pre.py: its helpers are pinned to the reference's;
    pycocotools / cv2 are not installable, see its header) with the reference's own np.random.choice draws (rng mode), the
    REFERENCE Net on the result, and the reference's result statements (run_inference_custom.py:290-307) -> tests/golden/frame.npz."""
    import json
    from . import pem_pre as opre
    du = rh.pem_data_utils()
    c = FRAME_CASE
    inp = _frame_inputs()
    ism = np.load(os.path.join(tmp, "ism.npz"))
    dets_ = json.load(open(os.path.join(tmp, "detection_ism.json")))
    dets = [d for d in dets_ if d["score"] > c["det_score_thresh"]]
    kept_ism = [i for i, d in enumerate(dets_) if d["score"] > c["det_score_thresh"]]
    masks = np.stack([du.rle_to_binary_mask(d["segmentation"]) for d in dets]).astype(bool)
    depth = inp["depth_mm"].numpy() * np.float32(inp["depth_scale"]) / np.float32(1000.0)
    rng = np.random.RandomState(c["rng_seed"])
    obs = opre.preprocess_frame(inp["rgb"], depth, inp["K"].numpy(), masks, inp["radius"], rng=rng)
    M = obs["pts"].shape[0]
    dense_fo = torch.randn(1, 2048, 256, generator=torch.Generator().manual_seed(c["feat_seed"])).expand(M, -1, -1).contiguous()
    ns = rh.pem()
    net = ns.pose_estimation_model.Net(rh.pem_cfg().model).eval()
    seeded.load_seeded(net, c["weight_seed"])
    ep = dict(pts=torch.from_numpy(obs["pts"]), rgb=torch.from_numpy(obs["rgb"]), rgb_choose=torch.from_numpy(obs["rgb_choose"]),
              model=torch.from_numpy(inp["model"])[None].expand(M, -1, -1).contiguous(),
              dense_po=torch.from_numpy(inp["dense_po"])[None].expand(M, -1, -1).contiguous(), dense_fo=dense_fo)
    with torch.no_grad():
        torch.manual_seed(c["rand_seed"])
        out = net(dict(ep))
    # the reference's result statements on its own outputs (pose score x detection score; metres -> millimetres; json records)
    pem_dir = os.path.join(rh.REF_ROOT, "SAM-6D", "Pose_Estimation_Model")
    src = _ref_statements(os.path.join(pem_dir, "run_inference_custom.py"), "if 'pred_pose_score' in out.keys():", "json.dump(detections, f)")
    sub = [dets[i] for i in obs["kept"].tolist()]
    import copy
    # `out` is the end_points dict of the reference: the detection scores ride through it (all_score, :239, :247)
    env = dict(out=dict({k: v.clone() for k, v in out.items() if torch.is_tensor(v)},
                        score=torch.FloatTensor([d["score"] for d in sub])),
               detections=copy.deepcopy(sub), json=json, os=os, np=np, torch=torch,
               cfg=types.SimpleNamespace(output_dir=tmp), open=open)
    os.makedirs(os.path.join(tmp, "sam6d_results"), exist_ok=True)
    exec(src, env)
    rec = {("ism_" + k): ism[k] for k in ism.files}
    rec.update(ism_json=np.array(json.dumps(dets_)), kept_ism=np.array(kept_ism), kept_pre=obs["kept"], pts=obs["pts"], rgb_choose=obs["rgb_choose"],
               bbox=obs["bbox"], pred_R=out["pred_R"].numpy(), pred_t=out["pred_t"].numpy(), pred_pose_score=out["pred_pose_score"].numpy(),
               pem_json=np.array(open(os.path.join(tmp, "sam6d_results", "detection_pem.json")).read()), case=np.array(str(c)))
    rec["rgb_sum"], rec["rgb_smp"] = digest(torch.from_numpy(obs["rgb"]), 4099)
    np.savez_compressed(os.path.join(OUT, "frame.npz"), **rec)
    print("frame.npz: ISM kept", kept_ism, "-> PEM kept", obs["kept"].tolist(), "pose score", rec["pred_pose_score"].round(4).tolist())


def gen_frame():
    """tests/golden/frame.npz -- ONE frame through the whole chain in the reference's order (VERDICT r2 missing #7): ISM scoring ->
    detection selection -> JSON hand-off -> score threshold -> per-detection pre-processing -> Net -> result records.  The two
    reference trees own the same top-level module names, so the stages run in two processes."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for stage in ("frame_ism", "frame_pem"):
            subprocess.check_call([sys.executable, "-m", "oracle.gen_golden", stage, tmp], cwd=os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])


def _samdec_ref(ns, cfg, seed):
    pe = ns.PromptEncoder(embed_dim=cfg["dim"], image_embedding_size=(cfg["emb"],) * 2,
                          input_image_size=(cfg["img"],) * 2, mask_in_chans=16)
    md = ns.MaskDecoder(num_multimask_outputs=cfg["n_multi"],
                        transformer=ns.TwoWayTransformer(depth=cfg["depth"], embedding_dim=cfg["dim"], mlp_dim=cfg["mlp"],
                                                         num_heads=cfg["heads"]),
                        transformer_dim=cfg["dim"], iou_head_depth=cfg["iou_depth"], iou_head_hidden_dim=cfg["iou_hidden"])
    m = torch.nn.Module()
    m.prompt_encoder, m.mask_decoder = pe, md
    return seeded.load_seeded(m.eval(), seed)


def gen_sam_decoder():
    """Reference PromptEncoder + MaskDecoder (+ TwoWayTransformer) run unmodified, as Sam.forward / SamPredictor
    call them; Sam.postprocess_masks is three lines of F.interpolate and is restated by the oracle."""
    from . import sam_decoder as od
    ns = rh.sam_decoder()
    c = SAMDEC_CASE
    rec = {}

    def run(m, emb, **kw):
        s, d = m.prompt_encoder(points=kw.get("points"), boxes=kw.get("boxes"), masks=None)
        mk, iou = m.mask_decoder(image_embeddings=emb, image_pe=m.prompt_encoder.get_dense_pe(),
                                 sparse_prompt_embeddings=s, dense_prompt_embeddings=d,
                                 multimask_output=kw.get("multi", True))
        return s, mk, iou
    with torch.no_grad():
        cfg = od.MINI
        m = _samdec_ref(ns, cfg, c["weight_seed"])
        inp = synth.sam_decoder_inputs(cfg, c["n_mini"], c["input_seed"])
        s, mk, iou = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
        rec["mini_sparse"], rec["mini_masks"], rec["mini_iou"] = s.numpy(), mk.numpy(), iou.numpy()
        rec["mini_dense_pe"] = m.prompt_encoder.get_dense_pe().numpy()
        s, mk, iou = run(m, inp["emb"], points=(inp["points2"], inp["labels2"]), multi=False)
        rec["mini_sparse2"], rec["mini_masks2"], rec["mini_iou2"] = s.numpy(), mk.numpy(), iou.numpy()
        s, mk, iou = run(m, inp["emb"], boxes=inp["boxes"])
        rec["mini_sparse_box"], rec["mini_masks_box"], rec["mini_iou_box"] = s.numpy(), mk.numpy(), iou.numpy()
        rec["mini_post"] = od.postprocess_masks(torch.from_numpy(rec["mini_masks"][:3]), cfg["img"], c["mini_input_size"],
                                                c["mini_orig"]).numpy()
        rec["mini_keys"] = np.array(sorted(m.state_dict().keys()))
        rec["mini_shapes"] = np.array([str(tuple(m.state_dict()[k].shape)) for k in sorted(m.state_dict().keys())])
        cfg = od.SAM
        m = _samdec_ref(ns, cfg, c["weight_seed"])
        inp = synth.sam_decoder_inputs(cfg, c["n_full"], c["input_seed"])
        s, mk, iou = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
        rec["sam_iou"] = iou.numpy()
        rec["sam_masks_sum"], rec["sam_masks_smp"] = digest(mk, 211)
        rec["sam_keys"] = np.array(sorted(m.state_dict().keys()))
        rec["sam_shapes"] = np.array([str(tuple(m.state_dict()[k].shape)) for k in sorted(m.state_dict().keys())])
        # mask post-processing of the automatic mask generator: reference amg.py functions on the upscaled logits
        low = synth.sam_lowres_logits(c["post_B"], 3, 256, c["post_seed"])
        full = od.postprocess_masks(low, 1024, c["post_input_size"], c["post_orig"]).flatten(0, 1)
        rec["post_stability"] = ns.amg.calculate_stability_score(full, 0.0, 1.0).numpy()
        mb = full > 0.0
        rec["post_boxes"] = ns.amg.batched_mask_to_box(mb).numpy()
        rec["post_area"] = mb.flatten(1).sum(1).numpy()
        rec["post_bits"] = np.packbits(mb.numpy().reshape(mb.shape[0], -1)[:, ::7], axis=1)
        # generator bookkeeping: the point grid and the resized-frame shape / prompt coordinates
        rec["grid32"] = ns.amg.build_point_grid(32)
        sizes = [(480, 640), (1080, 1920), (333, 499), (1024, 1024), (2000, 1500)]
        rls = ns.transforms.ResizeLongestSide(1024)
        rec["pre_shapes"] = np.array([rls.get_preprocess_shape(h, w, 1024) for h, w in sizes])
        rec["pre_sizes"] = np.array(sizes)
        rec["coords_480x640"] = rls.apply_coords(rec["grid32"] * np.array([[640, 480]]), (480, 640))
        # claim used by sam6d_amd.sam.amg.generate_proposals: with ONE crop spanning the frame the crop-edge filter of
        # _process_batch never fires and uncrop_boxes_xyxy is the identity -- checked here with the reference functions
        gg = torch.Generator().manual_seed(3)
        xy = (torch.rand(4000, 2, generator=gg) * torch.tensor([640.0, 480.0])).floor()
        bx = torch.cat([xy, (xy + torch.rand(4000, 2, generator=gg) * 300).minimum(torch.tensor([639.0, 479.0])).floor()], 1)
        bx = torch.cat([bx, torch.tensor([[0.0, 0, 639, 479], [0, 0, 10, 10], [630, 470, 639, 479], [19, 19, 621, 461]])])
        assert not ns.amg.is_box_near_crop_edge(bx, [0, 0, 640, 480], [0, 0, 640, 480]).any()
        assert torch.equal(ns.amg.uncrop_boxes_xyxy(bx, [0, 0, 640, 480]), bx)
    rec["case"] = np.array(str(c))
    np.savez_compressed(os.path.join(OUT, "sam_decoder.npz"), **rec)
    print("sam_decoder.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in rec.items()})
    print("mini iou", rec["mini_iou"][:2], "mask |max|", np.abs(rec["mini_masks"]).max(), "sam iou", rec["sam_iou"][:2])


def gen_dinov2():
    """Reference CustomDINOv2 methods (crop pipeline + masked patch features) and DinoVisionTransformer, run
    unmodified; only rgb_normalize (torchvision ToTensor + Normalize, un-vendored) is supplied by the oracle."""
    import importlib

    from . import dinov2 as od
    rh.ism()
    vt = importlib.import_module("model.vision_transformer")
    dv = importlib.import_module("model.dinov2")
    bu = importlib.import_module("utils.bbox_utils")
    c = DINO_CASE
    inp = synth.dinov2_inputs(P=c["P"], seed=c["input_seed"])

    def custom(model, target, cfg):
        o = object.__new__(dv.CustomDINOv2)               # the constructor needs a checkpoint file + torchvision
        torch.nn.Module.__init__(o)
        o.model, o.chunk_size, o.patch_size, o.proposal_size = model, 4, cfg["patch"], target
        o.validpatch_thresh, o.token_name = 0.5, "x_norm_clstoken"
        o.rgb_normalize = od.rgb_normalize
        o.rgb_proposal_processor = bu.CropResizePad(target)
        o.patch_kernel = torch.nn.AvgPool2d(kernel_size=cfg["patch"], stride=cfg["patch"])
        return o

    def props():
        return types.SimpleNamespace(masks=inp["masks"].clone(), boxes=inp["boxes"].clone())
    rec = {}
    with torch.no_grad():
        cfg = od.MINI
        m = vt.DinoVisionTransformer(img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["dim"],
                                     depth=cfg["depth"], num_heads=cfg["heads"], mlp_ratio=4, init_values=1.0,
                                     block_chunks=0).eval()
        seeded.load_seeded(m, c["weight_seed"])
        o = custom(m, c["mini_target"], cfg)
        rec["mini_rgbs"] = o.process_rgb_proposals(inp["image"], inp["masks"].clone(), inp["boxes"]).numpy()
        rec["mini_masks"] = o.process_masks_proposals(inp["masks"].clone(), inp["boxes"]).numpy()
        cls, patch = o.forward(inp["image"], props())
        rec["mini_cls"], rec["mini_patch"] = cls.numpy(), patch.numpy()
        rec["mini_cls_only"] = o.forward_cls_token(inp["image"], props()).numpy()
        rec["mini_patch_only"] = o.forward_patch_tokens(inp["image"], props()).numpy()
        rec["mini_keys"] = np.array(sorted(m.state_dict().keys()))
        # released configuration: ViT-L/14 at 224 (pos_embed interpolated 37x37 -> 16x16), full-size frame crops
        cfg = od.VIT_L14
        m = vt.vit_large(patch_size=14, img_size=518, init_values=1.0, ffn_layer="mlp", block_chunks=0).eval()
        seeded.load_seeded(m, c["weight_seed"])
        o = custom(m, 224, cfg)
        rgbs = o.process_rgb_proposals(inp["image"], inp["masks"].clone(), inp["boxes"])
        pm = o.process_masks_proposals(inp["masks"].clone(), inp["boxes"])
        rec["l_rgbs_sum"], rec["l_rgbs_smp"] = digest(rgbs, 1009)
        rec["l_masks_sum"], rec["l_masks_smp"] = digest(pm, 1009)
        cls, patch = o.compute_cls_and_patch_features(rgbs[: c["n_full"]], pm[: c["n_full"]])
        rec["l_cls"] = cls.numpy()
        rec["l_patch_sum"], rec["l_patch_smp"] = digest(patch, 53)
        rec["l_keys"] = np.array(sorted(m.state_dict().keys()))
        rec["l_shapes"] = np.array([str(tuple(m.state_dict()[k].shape)) for k in sorted(m.state_dict().keys())])
    rec["case"] = np.array(str(c))
    np.savez_compressed(os.path.join(OUT, "dinov2.npz"), **rec)
    print("dinov2.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in rec.items()})


E2E_CASE = dict(sam_seed=3, dino_seed=6, O=3, T=6, patch_stride=8, pred_iou_thresh=0.3086, stability_score_thresh=0.691,
                stability_score_offset=0.02, box_nms_thresh=1.5, points_per_batch=64, confidence_thresh=0.2, ism_seed=21)


def e2e_templates_and_extra(rgb, P=10):
    """Inputs of the e2e case that do not come out of a model: the ten deterministic depth-window proposals of tests/util.frame_
    inputs on the Example frame (they join SAM's proposals before the descriptor stage, and are the template crops) and eight
    ellipse proposals of the DINOv2 tests (template crops only).  Shared with the test."""
    from tests import util as tutil
    fi = tutil.frame_inputs(dict(FRAME_CASE, P=P))
    ell = synth.dinov2_inputs(P=8, seed=E2E_CASE["ism_seed"])
    return fi, ell


def gen_frame_e2e():
    """tests/golden/frame_e2e.npz -- the Example frame from PIXELS to scored detections through the reference's own modules
    (VERDICT r3 item 1b): SamPredictor.set_image (ResizeLongestSide + Sam.preprocess + ViT-H) -> CustomSamAutomaticMaskGenerator.
    generate_masks (prompt encoder, mask decoder, Sam.postprocess_masks, IoU / stability filters, box NMS) -> CustomDINOv2.forward
    (crops, ViT-L/14, cls + masked patch descriptors) -> Instance_Segmentation_Model scoring methods (semantic / appearance /
    geometric score, run_inference_custom.py:160-199), seeded weights everywhere.

    Seeded-random SAM weights give masks that span the frame (their logits are +-0.05 of texture around 0: every box IS the
    frame), so the generator's thresholds are set for these weights through its CONSTRUCTOR (pred_iou_thresh,
    stability_score_thresh; box_nms_thresh > 1 = no suppression, or one proposal would be left) and ``stability_score_offset``;
    nothing in the reference code is changed.  The ten deterministic depth-window proposals of the
    Example frame join SAM's before the descriptor stage so that crops of ordinary sizes go through it too.  Template descriptors
    (3 objects x 6 templates) are the reference descriptor model's own output for 18 crops of the frame, stored in fp16 with every
    8th patch (an input definition: both sides read the stored values).  torchvision is not installable: batched_nms / box_area
    come from oracle/sam_decoder.py (unpinned, as everywhere), ToTensor + Normalize from oracle/dinov2.py, Pillow's resize is real."""
    import importlib

    from PIL import Image

    from . import dinov2 as odino
    from . import sam_decoder as od
    c = E2E_CASE
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ns = rh.ism()
    import segment_anything
    import segment_anything.automatic_mask_generator as amg_mod
    import segment_anything.utils.transforms as tr_mod
    msam = importlib.import_module("model.sam")
    mu = importlib.import_module("model.utils")
    vt = importlib.import_module("model.vision_transformer")
    dv = importlib.import_module("model.dinov2")
    bu = importlib.import_module("utils.bbox_utils")

    def batched_nms(boxes, scores, idxs, iou_threshold):
        assert (idxs == 0).all()
        return od.nms(boxes.float(), scores, iou_threshold)
    box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])      # noqa: E731
    amg_mod.batched_nms, amg_mod.box_area, msam.batched_nms, msam.box_area = batched_nms, box_area, batched_nms, box_area
    tr_mod.to_pil_image = lambda a: Image.fromarray(a)
    tr_mod.resize = lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR)

    g = np.load(os.path.join(OUT, "example_frame.npz"))
    rgb = g["rgb"]
    rec = {}
    with torch.no_grad():
        # ---- segmentor ------------------------------------------------------------------------------------------------------
        sam = segment_anything.sam_model_registry["vit_h"]()
        seeded.load_seeded(sam.eval(), c["sam_seed"])
        gen = msam.CustomSamAutomaticMaskGenerator(sam, points_per_batch=c["points_per_batch"], stability_score_thresh=c["stability_score_thresh"],
                                                   pred_iou_thresh=c["pred_iou_thresh"], box_nms_thresh=c["box_nms_thresh"])
        gen.stability_score_offset = c["stability_score_offset"]
        # every candidate's predicted IoU / stability score on the way (for the margins of the two filters): _process_batch wrapped
        cand = dict(iou=[], stab=[])
        real_filter = amg_mod.MaskData.filter

        def spy(self, keep):
            if "stability_score" in self._stats and "boxes" not in self._stats and len(self["stability_score"]) == len(keep):
                cand["stab"].append(self["stability_score"].clone())
            elif "iou_preds" in self._stats and "stability_score" not in self._stats and "rles" not in self._stats:
                cand["iou"].append(self["iou_preds"].clone())
            return real_filter(self, keep)
        amg_mod.MaskData.filter = spy
        real_set = gen.predictor.set_image

        def set_image(*a, **k):                       # the embedding is dropped again by reset_image(): keep a copy for the digest
            real_set(*a, **k)
            cand["emb"] = gen.predictor.features.clone()
        gen.predictor.set_image = set_image
        try:
            det = gen.generate_masks(rgb)
        finally:
            amg_mod.MaskData.filter = real_filter
        rec["emb_sum"], rec["emb_smp"] = digest(cand["emb"], 1009)
        masks, boxes = det["masks"], det["boxes"]
        K = masks.shape[0]
        rec["sam_masks"] = np.packbits(masks.numpy().astype(bool).reshape(K, -1), axis=1)
        rec["sam_boxes"] = boxes.numpy()
        rec["cand_iou"] = torch.cat(cand["iou"]).numpy()
        rec["n_after_iou_filter"] = np.array(sum(len(x) for x in cand["stab"]))
        rec["cand_stab_after_iou"] = torch.cat(cand["stab"]).numpy()
        print("SAM proposals", K, "boxes", boxes[:4].tolist(), "candidates past the IoU filter", int(rec["n_after_iou_filter"]))
        # ---- descriptors ----------------------------------------------------------------------------------------------------
        fi, ell = e2e_templates_and_extra(rgb)
        q_masks = torch.cat([masks.float(), fi["masks"]])
        q_boxes = torch.cat([boxes.long(), fi["boxes"].long()])              # integer XYXY, as batched_mask_to_box returns them
        m = vt.vit_large(patch_size=14, img_size=518, init_values=1.0, ffn_layer="mlp", block_chunks=0).eval()
        seeded.load_seeded(m, c["dino_seed"])
        o = object.__new__(dv.CustomDINOv2)
        torch.nn.Module.__init__(o)
        o.model, o.chunk_size, o.patch_size, o.proposal_size = m, 8, 14, 224
        o.validpatch_thresh, o.token_name = 0.5, "x_norm_clstoken"
        o.rgb_normalize = odino.rgb_normalize
        o.rgb_proposal_processor = bu.CropResizePad(224)
        o.patch_kernel = torch.nn.AvgPool2d(kernel_size=14, stride=14)
        detections = mu.Detections({"masks": q_masks.clone(), "boxes": q_boxes.clone()})
        qry_cls, qry_patch = o.forward(rgb, detections)
        t_masks = torch.cat([fi["masks"], ell["masks"][:, :480, :640]])[:c["O"] * c["T"]]
        t_boxes = torch.cat([fi["boxes"].long(), ell["boxes"].long()])[:c["O"] * c["T"]]
        t_cls, t_patch = o.forward(rgb, types.SimpleNamespace(masks=t_masks.clone(), boxes=t_boxes.clone()))
        ref_cls = t_cls.view(c["O"], c["T"], -1).half()
        ref_patch = t_patch[:, ::c["patch_stride"]].reshape(c["O"], c["T"], -1, t_patch.shape[-1]).half()
        rec["ref_cls"], rec["ref_patch"] = ref_cls.numpy(), ref_patch.numpy()
        rec["qry_cls"] = qry_cls.numpy()
        rec["qry_patch_sum"], rec["qry_patch_smp"] = digest(qry_patch, 211)
        # ---- scoring (run_inference_custom.py:160-199) -----------------------------------------------------------------------
        d = synth.ism_inputs(P=4, O=c["O"], T=c["T"], C=8, n_patch=4, H=480, W=640, seed=c["ism_seed"])      # template poses only
        Det = ns.detector.Instance_Segmentation_Model
        fake = types.SimpleNamespace()
        pointcloud = fi["pointcloud"] * torch.tensor([1.0, 0.8, 1.2])[:c["O"]].view(-1, 1, 1)       # one model cloud per object
        fake.ref_data = dict(descriptors=ref_cls.float(), appe_descriptors=ref_patch.float(), poses=d["poses"], pointcloud=pointcloud)
        fake.matching_config = types.SimpleNamespace(metric=ns.loss.PairwiseSimilarity(), aggregation_function="avg_5",
                                                     confidence_thresh=c["confidence_thresh"])
        for name in ("best_template_pose", "compute_semantic_score", "compute_appearance_score", "compute_geometric_score",
                     "project_template_to_image", "Calculate_the_query_translation"):
            setattr(fake, name, types.MethodType(getattr(Det, name), fake))
        sel, pobj, sem, bt = fake.compute_semantic_score(qry_cls)
        detections.filter(sel)
        qp = qry_patch[sel, :]
        appe, ref = fake.compute_appearance_score(bt, pobj, qp)
        batch = dict(depth=[fi["depth_mm"]], cam_intrinsic=[fi["K"]], depth_scale=fi["depth_scale"])
        uv = fake.project_template_to_image(bt, pobj, batch, detections.masks)
        geo, vr = fake.compute_geometric_score(uv, detections, qp, ref, visible_thred=0.5)
        final = (sem + appe + geo * vr) / (1 + 1 + vr)
    rec.update(sel=sel.numpy(), pred_obj=pobj.numpy(), best_template=bt.numpy(), semantic=sem.numpy(), appearance=appe.numpy(),
               visible_ratio=vr.numpy(), iou=np.asarray(geo if not torch.is_tensor(geo) else geo.numpy(), dtype=np.float32) * np.ones(len(sel), np.float32),
               final=final.numpy(), image_uv=uv.numpy(), case=np.array(str(c)))
    np.savez_compressed(os.path.join(OUT, "frame_e2e.npz"), **rec)
    print("frame_e2e.npz: selected", sel.tolist(), "objects", pobj.tolist(), "templates", bt.tolist(), "final", final.numpy().round(4).tolist())
    print("semantic", sem.numpy().round(4).tolist())
    print("size", os.path.getsize(os.path.join(OUT, "frame_e2e.npz")))


E2E_NMS_CASE = dict(sam_seed=3, mask_threshold=0.18, pred_iou_thresh=0.2763, stability_score_thresh=0.1, stability_score_offset=0.02,
                    box_nms_thresh=0.7, points_per_batch=64)


def gen_frame_e2e_nms():
    """tests/golden/frame_e2e_nms.npz -- a second pixels-to-proposals case in which the generator's box NMS REALLY suppresses
    (VERDICT r4 missing #3 / next #1e; in frame_e2e.npz every box is the frame and box_nms_thresh is 1.5).  Same frame, same seeded
    SAM; what changes is the model's ``mask_threshold`` attribute (Sam.mask_threshold, modeling/sam.py:18: 0.18 instead of 0.0 -- an
    instance attribute, no reference code is touched): seeded mask logits are texture of +-0.08 around 0, so above 0.18 (at the frame's resolution the 99.9th percentile of the logits is 0.17) a mask
    is a sparse set of its own highest pixels and its box follows the prompt.  Thresholds by a rule on a low-resolution dry run
    (the 384 best predicted IoUs, stability >= 0.1 at offset 0.02; a dry run on the decoder's logits upsampled as postprocess_masks does): about ninety candidates reach batched_nms (model/sam.py:138-144) at the
    reference's own box_nms_thresh = 0.7 and about a tenth survive.  Stored: the inputs and the result of that NMS call (spied), and
    the proposals generate_masks returns.  torchvision is not installable: batched_nms is oracle/sam_decoder.py's (unpinned)."""
    import importlib

    from PIL import Image

    from . import sam_decoder as od
    c = E2E_NMS_CASE
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    rh.ism()
    import segment_anything
    import segment_anything.automatic_mask_generator as amg_mod
    import segment_anything.utils.transforms as tr_mod
    msam = importlib.import_module("model.sam")
    spy = {}

    def batched_nms(boxes, scores, idxs, iou_threshold):
        assert (idxs == 0).all()
        keep = od.nms(boxes.float(), scores, iou_threshold)
        spy.update(boxes=boxes.clone(), scores=scores.clone(), keep=keep.clone(), thr=iou_threshold)
        return keep
    box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])      # noqa: E731
    amg_mod.batched_nms, amg_mod.box_area, msam.batched_nms, msam.box_area = batched_nms, box_area, batched_nms, box_area
    tr_mod.to_pil_image = lambda a: Image.fromarray(a)
    tr_mod.resize = lambda im, size: im.resize((size[1], size[0]), Image.BILINEAR)
    rgb = np.load(os.path.join(OUT, "example_frame.npz"))["rgb"]
    with torch.no_grad():
        sam = segment_anything.sam_model_registry["vit_h"]()
        seeded.load_seeded(sam.eval(), c["sam_seed"])
        sam.mask_threshold = c["mask_threshold"]
        gen = msam.CustomSamAutomaticMaskGenerator(sam, points_per_batch=c["points_per_batch"], stability_score_thresh=c["stability_score_thresh"],
                                                   pred_iou_thresh=c["pred_iou_thresh"], box_nms_thresh=c["box_nms_thresh"])
        gen.stability_score_offset = c["stability_score_offset"]
        det = gen.generate_masks(rgb)
    masks, boxes = det["masks"], det["boxes"]
    K = masks.shape[0]
    rec = dict(masks=np.packbits(masks.numpy().astype(bool).reshape(K, -1), axis=1), boxes=boxes.numpy(), nms_in_boxes=spy["boxes"].numpy(),
               nms_in_scores=spy["scores"].numpy(), nms_keep=spy["keep"].numpy(), case=np.array(str(c)))
    np.savez_compressed(os.path.join(OUT, "frame_e2e_nms.npz"), **rec)
    print("frame_e2e_nms.npz: candidates into NMS", len(spy["scores"]), "kept", len(spy["keep"]), "proposals returned", K, "mask areas",
          masks.flatten(1).sum(1)[:10].tolist(), "size", os.path.getsize(os.path.join(OUT, "frame_e2e_nms.npz")))


def _e2e_query_proposals(g):
    """The proposals the scoring half of frame_e2e.npz scored: SAM's (stored bit-packed) followed by the ten depth windows."""
    from tests import util as tutil
    fi = tutil.frame_inputs(dict(FRAME_CASE, P=10))
    K = g["sam_boxes"].shape[0]
    sam_masks = torch.from_numpy(np.unpackbits(g["sam_masks"], axis=1)[:, :480 * 640].reshape(K, 480, 640).astype(np.float32))
    return fi, torch.cat([sam_masks, fi["masks"]]), torch.cat([torch.from_numpy(g["sam_boxes"]).long(), fi["boxes"].long()])


def gen_frame_e2e_ism(tmp):
    """Stage 1 of the pose half (ISM process): the REFERENCE's statements after the final score, on the reference's own scores
    stored in frame_e2e.npz -- two flows:
      bop    detector.py:349-352,383-400 (test_step): Detections.remove_very_small_detections -> [descriptors / scores: stored]
             -> filter -> add_attribute(scores / object_ids = pred_idx_objects) -> apply_nms_per_object_id(0.25) -> to_numpy ->
             save_to_file -> convert_npz_to_json;
      custom run_inference_custom.py:161-205 (demo.sh): no size filter, no NMS; object_ids keep the predicted objects (the
             script's zeros_like is its single-object special case: this case scores three objects).
    torchvision is not installable: box_area / nms are supplied as in gen_detections_ops (the NMS boundary stays unpinned)."""
    import importlib
    import json
    from tests import util as tutil
    from . import sam_decoder as od
    rh.ism()
    mu = importlib.import_module("model.utils")
    mu.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    mu.torchvision.ops.nms = od.nms
    pc = tutil.E2E_POSE_CASE
    g = np.load(os.path.join(OUT, "frame_e2e.npz"))
    fi, q_masks, q_boxes = _e2e_query_proposals(g)
    sel, final, pobj = torch.from_numpy(g["sel"]), torch.from_numpy(g["final"]), torch.from_numpy(g["pred_obj"])
    for flow in ("bop", "custom"):
        det = mu.Detections({"masks": q_masks.clone(), "boxes": q_boxes.clone()})
        if flow == "bop":
            n0 = len(det)
            det.remove_very_small_detections(types.SimpleNamespace(min_box_size=pc["min_box_size"], min_mask_size=pc["min_mask_size"]))
            assert len(det) == n0, "the stored scores were computed on the unfiltered proposal list"
        det.filter(sel)
        det.add_attribute("scores", final.clone())
        det.add_attribute("object_ids", pobj.clone())
        if flow == "bop":
            det.apply_nms_per_object_id(nms_thresh=pc["nms_thresh"])
        det.to_numpy()
        path = os.path.join(tmp, "det_" + flow)
        det.save_to_file(pc["scene_id"], pc["frame_id"], 0.0, path, pc["dataset"] if flow == "bop" else "Custom", return_results=False)
        recs = mu.convert_npz_to_json(idx=0, list_npz_paths=[path + ".npz"])
        json.dump(recs, open(path + ".json", "w"))
        print("frame_e2e_pose/ism", flow, len(recs), "detections, scores", [round(r["score"], 4) for r in recs])


def gen_frame_e2e_pem(tmp):
    """Stage 2 of the pose half (PEM process): template onboarding through the REFERENCE Net.feature_extraction.get_obj_feats
    (test_bop.py:117-119), the score threshold (bop_test_dataset.py:84 / run_inference_custom.py:169-171), the reference's rle
    decoder, the per-detection loop (oracle/pem_pre.py with injected sampling keys; one radius per detection = its object's,
    bop_test_dataset.py:125), per-object template rows (test_bop.py:143-147), the REFERENCE Net in one batch with the generator
    seeded (run_inference_custom.py:281-285), and the reference's two result writers executed from its files.  Detections are
    taken best-first (stable), the order sam6d_amd.pipeline.FramePipeline hands them on: a row of the batch does not depend on
    its neighbours in the reference either, the order only fixes which row of the injected randoms an instance consumes."""
    import copy
    import json
    import time
    from tests import util as tutil
    from . import pem_pre as opre
    du = rh.pem_data_utils()
    ns = rh.pem()
    pc = tutil.E2E_POSE_CASE
    pin = tutil.e2e_pose_inputs(pc)
    fi = pin["fi"]
    net = ns.pose_estimation_model.Net(rh.pem_cfg().model).eval()
    seeded.load_seeded(net, pc["pem_weight_seed"])
    pem_dir = os.path.join(rh.REF_ROOT, "SAM-6D", "Pose_Estimation_Model")
    rec = {}
    with torch.no_grad():
        dense_po, dense_fo = net.feature_extraction.get_obj_feats(pin["tem_rgb"], pin["tem_pts"], pin["tem_choose"])
    rec["dense_po"] = dense_po.numpy()
    for k in ("gt_R", "gt_t", "tem_obj_pts", "tem_v1"):                          # frozen: see tests/util.e2e_pose_inputs
        rec[k] = pin[k].numpy()
    rec["dense_fo_sum"], rec["dense_fo_smp"] = digest(dense_fo, 101)
    depth = fi["depth_mm"].numpy() * np.float32(fi["depth_scale"]) / np.float32(1000.0)
    for flow in ("bop", "custom"):
        dets_ = json.load(open(os.path.join(tmp, "det_" + flow + ".json")))
        order = sorted(range(len(dets_)), key=lambda i: -dets_[i]["score"])                 # stable: ties keep file order
        dets = [dets_[i] for i in order if dets_[i]["score"] > pc["det_score_thresh"]]
        masks = np.stack([du.rle_to_binary_mask(d["segmentation"]) for d in dets]).astype(bool)
        obj = np.array([d["category_id"] - 1 for d in dets])
        obs = opre.preprocess_frame(fi["rgb"], depth, fi["K"].numpy(), masks, pin["radius"].numpy()[obj], keys=pin["keys"].numpy()[:len(dets)])
        kept = obs["kept"]
        M = len(kept)
        o = torch.from_numpy(obj[kept])
        ep = dict(pts=torch.from_numpy(obs["pts"]), rgb=torch.from_numpy(obs["rgb"]), rgb_choose=torch.from_numpy(obs["rgb_choose"]),
                  model=pin["model"][o].contiguous(), dense_po=dense_po[o].contiguous(), dense_fo=dense_fo[o].contiguous())
        t0 = time.time()
        with torch.no_grad():
            torch.manual_seed(pc["rand_seed"])
            out = net(dict(ep))
        print("frame_e2e_pose/pem", flow, "reference Net on", M, "instances:", round(time.time() - t0, 1), "s")
        # ---- conditioning of the reference ITSELF.  compute_coarse_Rt (model_utils.py:187-246) draws 6000 hypotheses through a
        # searchsorted on a float32 cumsum of 38 416 weights, keeps 300 by a topk and picks one by an arg-max: three discontinuous
        # steps.  On unrelated features (seeded weights) near-ties exist, and the SAME torch code then answers differently on
        # another CPU (seen: the oracle's restatement on the GPU box's host against this container's).  Measured implementation
        # noise at the similarity matrix (product vs oracle, tools/probes/e2e_diag2.py): rms 1e-5 on values of rms 3.  The reference's
        # own compute_coarse_Rt is therefore called again 48 times per instance on its own inputs with that noise ADDED to the matrix
        # (the generator re-seeded to the same uniforms each time); an instance whose coarse pose moves in any trial cannot be held
        # to the bar by anybody: marked unstable, stored, reported by the tests instead of asserted.  The whole Net is also re-run
        # three times on inputs moved by 1e-5 x rms (colour crop, template features) for the continuous part.
        cap = {}
        cpm = ns.coarse_point_matching
        real_coarse = cpm.compute_coarse_Rt

        def spy(atten, pts1, pts2, *a, **k):
            cap.update(atten=atten.clone(), pts1=pts1.clone(), pts2=pts2.clone(), a=a, k=k)
            return real_coarse(atten, pts1, pts2, *a, **k)
        cpm.compute_coarse_Rt = spy
        try:
            with torch.no_grad():
                torch.manual_seed(pc["rand_seed"])
                chk = net(dict(ep))
        finally:
            cpm.compute_coarse_Rt = real_coarse
        assert torch.equal(chk["pred_R"], out["pred_R"])
        stable = np.ones(M, bool)
        coarse_move = np.zeros(M, np.float32)
        rms = cap["atten"].pow(2).mean().sqrt()
        alts = [[] for _ in range(M)]
        for trial in range(96):
            gq = torch.Generator().manual_seed(700 + trial)
            # even trials: noise on the similarity matrix; odd trials: the sparse points moved by float32 rounding-level relative
            # noise (2e-7) -- hypotheses from triplets with a repeated or collinear point have a rank-deficient 3 x 3 covariance
            # whose SVD completion is arbitrary (it follows the LAPACK build / CPU), they fit their own three points perfectly, so
            # they survive the topk, and on junk features one of them can win the arg-max
            noisy, q1, q2 = cap["atten"], cap["pts1"], cap["pts2"]
            if trial % 2 == 0:
                # (trials 48 .. 95, round 6: three times the noise -- a wider net for the set of hypotheses an implementation can land on)
                noisy = noisy + (1e-5 if trial < 48 else 3e-5) * torch.randn(noisy.shape, generator=gq)
            else:
                q1 = q1 * (1 + 2e-7 * torch.randn(q1.shape, generator=gq))
                q2 = q2 * (1 + 2e-7 * torch.randn(q2.shape, generator=gq))
            with torch.no_grad():
                torch.manual_seed(pc["rand_seed"])
                R2, t2 = real_coarse(noisy, q1, q2, *cap["a"], **cap["k"])
            d = (R2 - out["init_R"]).flatten(1).norm(dim=1).numpy()
            if trial < 48:                                   # `stable` keeps its round-5 definition (48 trials at the measured noise)
                coarse_move = np.maximum(coarse_move, d)
            # the hypotheses the reference itself lands on (round 6, VERDICT r5 next #2b): every coarse pose a trial produced that is
            # not one already seen for the instance -- its continuation through the fine stage is computed below, and the tests
            # hold an unstable instance to MEMBERSHIP in that set instead of only reporting it
            for i in range(M):
                seen = [out["init_R"][i]] + [a[0] for a in alts[i]]
                if all(float((R2[i] - s).norm()) > 1e-4 for s in seen) and len(alts[i]) < 6:
                    alts[i].append((R2[i].clone(), t2[i].clone()))
        stable &= coarse_move <= 1e-4
        K_alt = max(len(a) for a in alts)
        alt = dict(init_R=np.zeros((K_alt, M, 3, 3), np.float32), init_t=np.zeros((K_alt, M, 3), np.float32),
                   pred_R=np.zeros((K_alt, M, 3, 3), np.float32), pred_t=np.zeros((K_alt, M, 3), np.float32),
                   score=np.zeros((K_alt, M), np.float32), valid=np.zeros((K_alt, M), bool))
        for k in range(K_alt):
            R_ov, t_ov = out["init_R"].clone(), out["init_t"].clone()
            for i in range(M):
                if len(alts[i]) > k:
                    R_ov[i], t_ov[i] = alts[i][k]
                    alt["valid"][k, i] = True
            cpm.compute_coarse_Rt = lambda *a, **kw: (R_ov.clone(), t_ov.clone())
            try:
                with torch.no_grad():
                    torch.manual_seed(pc["rand_seed"])
                    o3 = net(dict(ep))
            finally:
                cpm.compute_coarse_Rt = real_coarse
            alt["init_R"][k], alt["init_t"][k] = R_ov.numpy(), t_ov.numpy()
            alt["pred_R"][k], alt["pred_t"][k], alt["score"][k] = o3["pred_R"].numpy(), o3["pred_t"].numpy(), o3["pred_pose_score"].numpy()
        print("  alternative coarse hypotheses per instance (reference under its own noise):", [len(a) for a in alts])
        print("  coarse pose movement under 1e-5 noise on the similarity matrix (rms", float(rms), ") / 2e-7 relative on the points, max over 48 trials:",
              coarse_move.round(4).tolist())
        move = np.zeros((3, M), np.float32)
        for trial in range(3):
            gq = torch.Generator().manual_seed(900 + trial)
            ep2 = dict(ep)
            for k in ("rgb", "dense_fo"):
                ep2[k] = ep[k] + 1e-5 * ep[k].pow(2).mean().sqrt() * torch.randn(ep[k].shape, generator=gq)
            with torch.no_grad():
                torch.manual_seed(pc["rand_seed"])
                o2 = net(ep2)
            dR2 = (o2["pred_R"] - out["pred_R"]).flatten(1).norm(dim=1).numpy()
            dt2 = (o2["pred_t"] - out["pred_t"]).norm(dim=1).numpy() * 1e3
            move[trial] = dR2
            stable &= (dR2 <= 1e-3) & (dt2 <= 1e-3)
        print("  reference pose movement under 1e-5 x rms additive input noise (max dR over three trials):", move.max(0).round(6).tolist(), "stable", stable.tolist())
        sub = [dets[i] for i in kept.tolist()]
        det_score = torch.FloatTensor([d["score"] for d in sub])
        p = flow + "_"
        # test_bop.py:155-181 (csv) and run_inference_custom.py:290-307 (json), executed from the reference's files
        src = _ref_statements(os.path.join(pem_dir, "test_bop.py"), "pred_Rs = torch.cat(pred_Rs", "image_time = time.time() - end")
        src2 = _ref_statements(os.path.join(pem_dir, "test_bop.py"), "# write results", "lines.append(line)")
        env = dict(torch=torch, time=time, end=time.time(), pred_Rs=[out["pred_R"]], pred_Ts=[out["pred_t"]], pred_scores=[out["pred_pose_score"]],
                   data=dict(score=det_score.reshape(1, M, 1), scene_id=torch.tensor([pc["scene_id"]]), img_id=torch.tensor([pc["frame_id"]]),
                             seg_time=torch.tensor([0.0]), obj_id=torch.tensor([d["category_id"] for d in sub]).reshape(1, M)),
                   n_instance=M, lines=[])
        exec(src, env)
        env["image_time"] = 0.0
        exec(src2, env)
        src3 = _ref_statements(os.path.join(pem_dir, "run_inference_custom.py"), "if 'pred_pose_score' in out.keys():", "json.dump(detections, f)")
        env3 = dict(out=dict({k: v.clone() for k, v in out.items() if torch.is_tensor(v)}, score=det_score),
                    detections=copy.deepcopy(sub), json=json, os=os, np=np, torch=torch, cfg=types.SimpleNamespace(output_dir=os.path.join(tmp, flow)), open=open)
        os.makedirs(os.path.join(tmp, flow, "sam6d_results"), exist_ok=True)
        exec(src3, env3)
        rec.update({p + "ism_json": np.array(json.dumps(dets_)), p + "order": np.array(order), p + "n_thresh": np.array(len(dets)), p + "kept_pre": kept,
                    p + "obj": obj[kept], p + "pts": obs["pts"], p + "rgb_choose": obs["rgb_choose"], p + "bbox": obs["bbox"],
                    p + "pred_R": out["pred_R"].numpy(), p + "pred_t": out["pred_t"].numpy(), p + "pred_pose_score": out["pred_pose_score"].numpy(),
                    p + "init_R": out["init_R"].numpy(), p + "init_t": out["init_t"].numpy(), p + "stable": stable,
                    p + "alt_init_R": alt["init_R"], p + "alt_init_t": alt["init_t"], p + "alt_pred_R": alt["pred_R"],
                    p + "alt_pred_t": alt["pred_t"], p + "alt_pred_pose_score": alt["score"], p + "alt_valid": alt["valid"], p + "ref_move_dR": move, p + "ref_coarse_move": coarse_move,
                    p + "csv": np.array("".join(env["lines"])),
                    p + "pem_json": np.array(open(os.path.join(tmp, flow, "sam6d_results", "detection_pem.json")).read())})
        rec[p + "rgb_sum"], rec[p + "rgb_smp"] = digest(torch.from_numpy(obs["rgb"]), 4099)
        # known answers: a detection whose mask IS the window its object was made from (tests/util.e2e_pose_inputs) observes that
        # object at the seeded pose
        kat = np.full(M, -1)
        for j, i in enumerate(kept.tolist()):
            for oo, wnd in enumerate(pc["base_windows"]):
                if obj[i] == oo and np.array_equal(masks[i], fi["masks"].numpy()[wnd] > 0):
                    kat[j] = oo
        rec[p + "kat_obj"] = kat
        for j in np.nonzero(kat >= 0)[0]:
            oo = kat[j]
            print("  known answer: instance", j, "object", oo, "|R - R0|", float((out["pred_R"][j] - pin["gt_R"][oo]).norm()),
                  "|t - t0| mm", float((out["pred_t"][j] - pin["gt_t"][oo]).norm() * 1e3), "pose score", float(out["pred_pose_score"][j]))
        print("  thresholded", len(dets), "of", len(dets_), "-> PEM kept", kept.tolist(), "objects", obj[kept].tolist())
        print("  pose score", out["pred_pose_score"].numpy().round(4).tolist())
        print("  t (m)", out["pred_t"].numpy().round(4).tolist())
    rec["case"] = np.array(str(pc))
    np.savez_compressed(os.path.join(OUT, "frame_e2e_pose.npz"), **rec)
    print("frame_e2e_pose.npz", os.path.getsize(os.path.join(OUT, "frame_e2e_pose.npz")), "bytes")


def gen_frame_e2e_pose():
    """tests/golden/frame_e2e_pose.npz -- the pose half of the pixels-to-pose golden: continues
    frame_e2e.npz (reference SAM + DINOv2 + scoring on the Example frame) through the reference's hand-off, the PEM's score
    threshold, pre-processing, onboarding and Net to R, t, pose score, BOP csv lines and detection_pem.json.  Two processes, as
    the two reference trees own the same top-level module names."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(OUT))
    with tempfile.TemporaryDirectory() as tmp:
        for stage in ("frame_e2e_ism", "frame_e2e_pem"):
            subprocess.check_call([sys.executable, "-m", "oracle.gen_golden", stage, tmp], cwd=root)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if sys.argv[1] in ("frame_ism", "frame_pem", "frame_e2e_ism", "frame_e2e_pem"):
        {"frame_ism": gen_frame_ism, "frame_pem": gen_frame_pem, "frame_e2e_ism": gen_frame_e2e_ism, "frame_e2e_pem": gen_frame_e2e_pem}[sys.argv[1]](sys.argv[2])
        sys.exit(0)
    {"frame": gen_frame, "frame_e2e": gen_frame_e2e, "frame_e2e_pose": gen_frame_e2e_pose, "frame_e2e_nms": gen_frame_e2e_nms, "pem": gen_pem, "pem_b32": gen_pem_b32, "pem_wc": gen_pem_wc, "example": gen_example, "sam": gen_sam, "ism": gen_ism, "dinov2": gen_dinov2, "sam_decoder": gen_sam_decoder, "handoff": gen_handoff, "pem_pre": gen_pem_pre, "pem_results": gen_pem_results, "detections_ops": gen_detections_ops, "sam_transforms": gen_sam_transforms, "sam_state_dict": gen_sam_state_dict}[sys.argv[1]]()
