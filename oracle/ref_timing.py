"""Time the REFERENCE's own modules on the host cores (test / measurement infrastructure: bench.py's ``cpu_baseline`` leg calls this
when /root/reference exists -- the build container; the GPU box has no reference tree and times the oracle port instead).

    python -m oracle.ref_timing sam|ism|pem [threads]      -> one JSON line {"leg", "seconds", "runs", "unit_count"}

The three legs are the three stages of one frame of BASELINE configs[1] with seeded weights through the reference's constructors
(SURVEY.md 8d): the SAM ImageEncoderViT (ViT-H) on one 1024 x 1024 frame; the ISM scoring methods on one frame of P = 128
proposals x 42 templates; the PEM Net on a batch of 8 instances (torch.manual_seed for its internal uniforms).  1 warm-up + the
median of 3 runs each.  Separate processes because the two reference trees own the same top-level module names."""
import json
import statistics
import sys
import time
import types

import torch

from sam6d_amd.utils import seeded, synth

from . import refharness as rh


def med3(fn):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def leg_sam():
    mod = rh.sam_builder()
    sam = mod.sam_model_registry["vit_h"]()
    enc = seeded.load_seeded(sam.image_encoder.eval(), 3)
    x = synth.sam_input(1, 5, 1024)
    return med3(lambda: enc(x)), 1


def leg_ism(P=128):
    ns = rh.ism()
    inp = synth.ism_inputs(P=P, O=1, T=42, seed=11)
    Det = ns.detector.Instance_Segmentation_Model
    fake = types.SimpleNamespace()
    fake.ref_data = dict(descriptors=inp["ref_cls"], appe_descriptors=inp["ref_patch"], poses=inp["poses"], pointcloud=inp["pointcloud"])
    fake.matching_config = types.SimpleNamespace(metric=ns.loss.PairwiseSimilarity(), aggregation_function="avg_5", confidence_thresh=0.2)
    for name in ("best_template_pose", "compute_semantic_score", "compute_appearance_score", "compute_geometric_score",
                 "project_template_to_image", "Calculate_the_query_translation"):
        setattr(fake, name, types.MethodType(getattr(Det, name), fake))

    def frame():
        sel, pobj, sem, bt = fake.compute_semantic_score(inp["qry_cls"])
        qp = inp["qry_patch"][sel]
        appe, ref = fake.compute_appearance_score(bt, pobj, qp)
        batch = dict(depth=[inp["depth"]], cam_intrinsic=[inp["K"]], depth_scale=1.0)
        uv = fake.project_template_to_image(bt, pobj, batch, inp["masks"][sel].clone())
        geo, vr = fake.compute_geometric_score(uv, types.SimpleNamespace(boxes=inp["boxes"][sel]), qp, ref, visible_thred=0.5)
        return (sem + appe + geo * vr) / (1 + 1 + vr)
    return med3(frame), 1


def leg_pem(B=8):
    ns = rh.pem()
    net = seeded.load_seeded(ns.pose_estimation_model.Net(rh.pem_cfg().model).eval(), 1)
    inp = synth.pem_inputs(B, seed=1)
    ep = {k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}

    def run():
        torch.manual_seed(2)
        return net(dict(ep))
    return med3(run), B


def main():
    leg = sys.argv[1]
    if len(sys.argv) > 2:
        torch.set_num_threads(int(sys.argv[2]))
    with torch.no_grad():
        (t, runs), n = {"sam": leg_sam, "ism": leg_ism, "pem": leg_pem}[leg]()
    print(json.dumps(dict(leg=leg, seconds=t, runs=runs, unit_count=n, threads=torch.get_num_threads())))


if __name__ == "__main__":
    main()
