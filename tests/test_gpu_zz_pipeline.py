"""One frame through the whole chain (sam6d_amd.pipeline.FramePipeline) with small seeded models: every stage hands
tensors to the next one and the result is a set of rigid poses."""
from functools import partial

import pytest
import torch

from oracle import dinov2 as odino
from oracle import sam as osam
from sam6d_amd.utils import seeded, synth

pytestmark = pytest.mark.gpu


def test_frame_through_all_five_models():
    run_frame(torch.device("cuda", 0))


def build_mini(dev, top_k=4, sync_stages=True):
    """FramePipeline on five small seeded models + one synthetic 120 x 160 frame -> (pipe, (img, depth, K, keys, rand_u))."""
    from sam6d_amd import pipeline
    from sam6d_amd.ism import dinov2 as pd
    from sam6d_amd.ism.scoring import FrameScorer
    from sam6d_amd.pem import pose_estimation_model as pm
    from sam6d_amd.sam import mask_decoder as md
    from sam6d_amd.sam.image_encoder import ImageEncoderViT
    c = osam.MINI                                                   # 512 px, 32 x 32 x 64 embedding
    enc = seeded.load_seeded(ImageEncoderViT(
        depth=c["depth"], embed_dim=c["dim"], img_size=c["img_size"], mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
        num_heads=c["heads"], patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=c["global_idx"],
        window_size=c["window"], out_chans=c["out_chans"]).eval(), 3).to(dev)
    dec = torch.nn.Module()
    dec.prompt_encoder = md.PromptEncoder(embed_dim=64, image_embedding_size=(32, 32), input_image_size=(512, 512), mask_in_chans=16)
    dec.mask_decoder = md.MaskDecoder(num_multimask_outputs=3, transformer=md.TwoWayTransformer(depth=2, embedding_dim=64, mlp_dim=96, num_heads=4),
                                      transformer_dim=64, iou_head_depth=3, iou_head_hidden_dim=48)
    dec = seeded.load_seeded(dec.eval(), 2).to(dev)
    dc = odino.MINI
    dino = pd.CustomDINOv2.__new__(pd.CustomDINOv2)
    torch.nn.Module.__init__(dino)
    dino.model = seeded.load_seeded(pd.DinoVisionTransformer(img_size=dc["img_size"], patch_size=14, embed_dim=dc["dim"], depth=dc["depth"],
                                                             num_heads=dc["heads"], mlp_ratio=4, init_values=1.0, block_chunks=0).eval(), 6).to(dev)
    dino.patch_size, dino.validpatch_thresh, dino.chunk_size, dino.proposal_size, dino.token_name = 14, 0.5, 64, 56, "x_norm_clstoken"
    ism = synth.ism_inputs(P=8, O=2, T=6, C=dc["dim"], n_patch=16, H=120, W=160, seed=4)
    scorer = FrameScorer(ism["ref_cls"].to(dev), ism["ref_patch"].to(dev), ism["poses"].to(dev), ism["pointcloud"].to(dev),
                         confidence_thresh=-1.0)
    pem = seeded.load_seeded(pm.Net(pm.default_cfg()).eval(), 1).to(dev)
    pin = synth.pem_inputs(1, seed=1)
    tpl = {k: pin[k].to(dev) for k in ("model", "dense_po", "dense_fo")}
    H, W = 120, 160
    g = torch.Generator().manual_seed(0)
    img = (torch.rand(H // 8, W // 8, 3, generator=g) * 255).repeat_interleave(8, 0).repeat_interleave(8, 1).to(torch.uint8).to(dev)
    depth = (0.8 + 0.05 * torch.rand(H, W, generator=g)).to(dev)
    K = torch.tensor([[143.0, 0, 80.0], [0, 143.0, 60.0], [0, 0, 1]], dtype=torch.float64).to(dev)
    keys = torch.rand(4, H * W, generator=g).to(dev)
    rand_u = synth.coarse_uniforms(4, 2).to(dev)
    pipe = pipeline.FramePipeline(enc, dec.prompt_encoder, dec.mask_decoder, dino, scorer, pem, tpl, object_radius=10.0, top_k=top_k,
                                  points_per_batch=8, min_box_size=0.0, min_mask_size=0.0,
                                  segmentor=dict(points_per_side=4, pred_iou_thresh=0.0, stability_score_thresh=0.0,
                                                 box_nms_thresh=1.0), sync_stages=sync_stages)
    return pipe, (img, depth, K, keys, rand_u)


def run_frame(dev, group_check=True):
    """The body, parametrised by the device so that tests/test_emu_pipeline.py can run it on the emulator (group_check=False there
    unless asked for: four more frames are most of an hour on the emulator)."""
    pipe, (img, depth, K, keys, rand_u) = build_mini(dev)
    H, W = img.shape[:2]
    det, poses = pipe(img, depth, K, keys, rand_u)
    assert set(pipe.times) >= {"sam_encoder", "proposals"}
    n = det.masks.shape[0]
    assert 1 <= n <= 4 and det.masks.shape[1:] == (H, W) and det.boxes.shape == (n, 4) and det.scores.shape == (n,)
    assert (det.scores[:-1] >= det.scores[1:]).all() and torch.isfinite(det.scores).all()
    assert poses is not None, "no detection survived the PEM pre-processing"
    M = poses["pred_R"].shape[0]
    assert 1 <= M <= n and poses["pred_t"].shape == (M, 3) and poses["kept"].shape == (M,)
    R = poses["pred_R"].double()
    eye = torch.eye(3, dtype=torch.float64, device=dev).expand(M, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-4) and torch.allclose(torch.linalg.det(R), torch.ones(M, dtype=torch.float64, device=dev), atol=1e-4)
    assert torch.isfinite(poses["pred_t"]).all() and torch.isfinite(poses["pred_pose_score"]).all()
    if not group_check:
        return
    # ---- a group of frames: one encoder pass and one PEM pass for the group, the same detections and poses as frame-by-frame calls
    import os
    img2 = img.flip(1).contiguous()
    depth2 = (depth + 0.02).contiguous()
    keys2, ru2 = keys.flip(0).contiguous(), rand_u.flip(0).contiguous()
    # BIT-IDENTICAL with the benched extractor (S6D_PEM_VIT_DTYPE=fp16: every GEMM of the PEM is this library's own, and a row of a
    # batch depends on its neighbours in none of its kernels; since round 5 the last library chains whose configuration followed the
    # batch size -- weighted_procrustes' sums / bmm and the (p - t) R product -- are fixed-order code).  With the fp32 extractor the
    # ViT-B's GEMMs are rocBLAS calls whose kernel choice follows the row count: measured 1.2e-6 in pred_R between a group of two
    # frames and the frames alone (tools/probes/group_exact.py, profiles/r05_group_exact.txt) and, after the encoder's window kernel
    # changed the masks' low bits, 2.3e-5 for one instance of this frame (the pose solver amplifies the features' last-bit noise by the
    # instance's conditioning) -- so that mode is held to the bar the reference-golden pose tests use for stable instances: 1e-3 in
    # R, 1e-3 mm in t (t is in metres here), the score to 1e-5.
    modes = (("fp32", False),) if dev.type == "cpu" else (("fp16", True), ("fp32", False))
    old = os.environ.get("S6D_PEM_VIT_DTYPE")
    try:
        for dt, exact in modes:
            os.environ["S6D_PEM_VIT_DTYPE"] = dt; __import__("sam6d_amd.policy").policy.reload()
            same = torch.equal if exact else (lambda a, b: torch.allclose(a, b, rtol=0, atol=1e-5))
            same_R = torch.equal if exact else (lambda a, b: torch.allclose(a, b, rtol=0, atol=1e-3))
            same_t = torch.equal if exact else (lambda a, b: torch.allclose(a, b, rtol=0, atol=1e-6))
            single = [pipe(img, depth, K, keys, rand_u), pipe(img2, depth2, K, keys2, ru2)]
            group = pipe.run_group([(img, depth, K, keys, rand_u), (img2, depth2, K, keys2, ru2)])
            assert len(group) == 2
            for (d1, p1), (d2, p2) in zip(single, group):
                assert torch.equal(d1.masks, d2.masks) and torch.equal(d1.boxes, d2.boxes) and torch.equal(d1.object_ids, d2.object_ids)
                assert torch.equal(d1.scores, d2.scores)
                assert (p1 is None) == (p2 is None)
                if p1 is not None:
                    assert torch.equal(p1["kept"], p2["kept"])
                    assert same_R(p1["pred_R"], p2["pred_R"]) and same_t(p1["pred_t"], p2["pred_t"]), dt
                    assert same(p1["pred_pose_score"], p2["pred_pose_score"]), dt
    finally:
        os.environ.pop("S6D_PEM_VIT_DTYPE") if old is None else os.environ.__setitem__("S6D_PEM_VIT_DTYPE", old); __import__("sam6d_amd.policy").policy.reload()


def mini_frames(frame, n=4):
    """n frames of one size made from the mini frame (flips / rolls / depth offsets), each with its own injected randoms."""
    img, depth, K, keys, rand_u = frame
    out = []
    for i in range(n):
        im = img if i == 0 else (img.flip(1) if i == 1 else (img.flip(0) if i == 2 else img.roll(24 * i, 1)))
        out.append((im.contiguous(), (depth + 0.01 * i).contiguous(), K, keys.roll(i, 0).contiguous(), rand_u.roll(i, 0).contiguous()))
    return out
