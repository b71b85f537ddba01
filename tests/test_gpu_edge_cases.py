"""Edge cases and BASELINE-size property checks on the GPU (empty / ragged inputs, maximum sizes, collisions)."""
import numpy as np
import pytest
import torch

from oracle import ism as oism
from oracle import pn2 as opn2
from sam6d_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from sam6d_amd import ops
    return ops


def test_empty_batches_are_no_ops(ops):
    dev = "cuda"
    assert ops.furthest_point_sampling(torch.zeros(0, 16, 3, device=dev), 4).shape == (0, 4)
    assert ops.ball_query(torch.zeros(0, 5, 3, device=dev), torch.zeros(0, 9, 3, device=dev), 0.1, 8).shape == (0, 5, 8)
    assert ops.gather_rows(torch.zeros(2, 7, 4, device=dev), torch.zeros(2, 0, dtype=torch.int32, device=dev)).shape == (2, 0, 4)
    assert ops.pairwise_cosine(torch.zeros(0, 1024, device=dev), torch.zeros(5, 1024, device=dev)).shape == (0, 5)
    a, r = ops.patch_scores(torch.zeros(0, 256, 1024, device=dev), torch.zeros(1, 1, 256, 1024, device=dev),
                            torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev), 0.5)
    assert a.shape == (0,) and r.shape == (0,)


def test_fps_all_points_and_single_point(ops):
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 130, 3, generator=g)
    idx = ops.furthest_point_sampling(x.cuda(), 130).cpu()           # M == N: a permutation
    assert torch.equal(idx, opn2.furthest_point_sampling(x, 130))
    assert all(sorted(r.tolist()) == list(range(130)) for r in idx)
    assert torch.equal(ops.furthest_point_sampling(x[:, :1].contiguous().cuda(), 1).cpu(), torch.zeros(2, 1, dtype=torch.int32))


def test_ball_query_nsample_larger_than_cloud_and_coincident_points(ops):
    x = torch.zeros(1, 40, 3)                                        # every point coincides: all in range
    x[0, 20:] = 5.0
    out = ops.ball_query(x.cuda(), x.cuda(), 0.1, 64).cpu()
    assert torch.equal(out, opn2.ball_query(x, x, 0.1, 64))
    assert (out[0, 0, :20] == torch.arange(20)).all() and (out[0, 0, 20:] == 0).all()     # first-hit fill


def test_template_onboarding_size_fps(ops):
    """210000 -> 2048 (42 views x 5000 px, feature_extraction.py:170-181): bit-exact at the maximum size."""
    g = torch.Generator().manual_seed(2)
    x = torch.rand(1, 210000, 3, generator=g)
    assert torch.equal(ops.furthest_point_sampling(x.cuda(), 2048).cpu(), opn2.furthest_point_sampling(x, 2048))


def test_tless_size_scoring_bit_identical_to_the_reference_run():
    """BASELINE configs[3]: T-LESS, 30 objects x 42 templates, many proposals (P = 256).  Held to the REFERENCE's own run at this
    size (tests/golden/ism_scoring_tless.npz, made by oracle/gen_golden.py from Instance_Segmentation_Model/model/detector.py:
    198-322): every integer output -- sel, pred_obj, best_template, the projected pixels, the boxes -- bit for bit, the query
    translation bit for bit, scores to float rounding.  The host oracle (same inputs) must agree with both."""
    import ast

    from sam6d_amd.ism.scoring import FrameScorer
    from tests import util
    g = util.golden("ism_scoring_tless.npz")
    c = ast.literal_eval(str(g["case"]))
    assert (c["P"], c["O"], c["T"]) == (256, 30, 42)
    inp = synth.ism_inputs(P=c["P"], O=c["O"], T=c["T"], seed=c["seed"])
    ref = oism.score_frame(inp)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    fs = FrameScorer(dev["ref_cls"], dev["ref_patch"], dev["poses"], dev["pointcloud"])
    out = fs.score(dev["qry_cls"], dev["qry_patch"], dev["masks"], dev["boxes"], dev["depth"], dev["K"])
    for k in ("sel", "pred_obj", "best_template"):
        assert np.array_equal(out[k].cpu().numpy(), g[k]), k
        assert torch.equal(out[k].cpu().long(), ref[k].long()), k
    assert np.array_equal(out["image_uv"].cpu().numpy(), g["image_uv"])
    t = fs.Calculate_the_query_translation(dev["masks"][out["sel"]].clone(), dev["depth"], dev["K"], 1.0).cpu().numpy()
    assert t.dtype == np.float32 and np.array_equal(t, g["translation"])
    box = torch.cat((out["image_uv"].min(1).values, out["image_uv"].max(1).values), -1).cpu().numpy()
    assert np.array_equal(box, np.concatenate((g["image_uv"].min(1), g["image_uv"].max(1)), -1))
    for k, tol in (("semantic", 1e-5), ("appearance", 1e-5), ("visible_ratio", 1e-5), ("final", 1e-5)):
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=0, atol=tol, err_msg=k)
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=tol, err_msg=k + " (oracle)")
    n = len(g["sel"])
    iou = torch.as_tensor(out["iou"]).cpu().numpy() * np.ones(n, np.float32)
    np.testing.assert_allclose(iou, g["iou"] * np.ones(n, np.float32), rtol=0, atol=1e-6)


def test_pem_batch32_properties():
    """BASELINE configs[1] size (B = 32): rotations are proper, scores in [0,1], known answer recovered."""
    from sam6d_amd.pem import pose_estimation_model as pm
    from sam6d_amd.utils import seeded
    net = seeded.load_seeded(pm.Net(pm.default_cfg()).eval(), 1).cuda()
    B = 32
    inp = synth.pem_inputs(B, seed=123, with_rgb=False)
    radius = torch.norm(inp["dense_po"], dim=2).max(1)[0]
    s = radius.reshape(-1, 1, 1) + 1e-6
    ep = dict(model=inp["model"].cuda(), coarse_rand_u=synth.coarse_uniforms(B, 9).cuda())
    with torch.no_grad():
        out = net.match((inp["pts"] / s).cuda(), inp["dense_fm_kat"].cuda(), (inp["dense_po"] / s).cuda(),
                        inp["dense_fo"].cuda(), radius.cuda(), ep)
    R = out["pred_R"].cpu()
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(B, 3, 3), atol=1e-5)
    assert torch.allclose(torch.det(R), torch.ones(B), atol=1e-5)
    assert (R - inp["gt_R"]).norm(dim=(1, 2)).max() < 1e-3
    assert (out["pred_t"].cpu() - inp["gt_t"]).abs().max() < 1e-4
    sc = out["pred_pose_score"].cpu()
    assert (sc >= 0).all() and (sc <= 1).all()


def test_next_row_entry_points_reject_bad_arguments_and_accept_empty_batches(ops):
    """C-ABI argument checks of the section-8f kernels: empty batches are no-ops, malformed shapes are refused with
    S6D_EINVAL (a RuntimeError on the Python side), nothing is launched."""
    import ctypes

    from sam6d_amd import _lib
    L = _lib.lib()
    null = ctypes.c_void_p(0)
    f3 = (ctypes.c_float * 3)(0, 0, 0)
    # B = 0: OK without touching any pointer
    assert L.s6d_crop_resize_pad_f32(null, null, null, 0, 480, 640, 224, f3, f3, null, null, null) == 0
    assert L.s6d_samdec_img2tok_bf16(null, null, null, null, null, null, null, null, ctypes.c_float(1e-5), 0, 4096, 7, 128, 0,
                                     0, null, null) == 0
    assert L.s6d_samdec_tok2img_f32(null, null, 384, 0, 128, 0, null, 0, 4096, ctypes.c_float(0.25), null, null) == 0
    assert L.s6d_samdec_upscale_heads_bf16(null, null, null, ctypes.c_float(1e-6), null, null, null, 0, 4, 64, 64, 256, null,
                                           null) == 0
    assert L.s6d_sam_mask_post_f32(null, 0, 256, 1024, 768, 1024, 480, 640, ctypes.c_float(0), ctypes.c_float(1), null, null,
                                   null) == 0
    # malformed: more than 8 prompt tokens, token count not a multiple of 16, k/v windows outside the row, crop
    # target 0, valid region larger than the padded square, NULL operands with B > 0
    bad = [
        L.s6d_samdec_img2tok_bf16(null, null, null, null, null, null, null, null, ctypes.c_float(1e-5), 1, 4096, 9, 128, 0, 0,
                                  null, null),
        L.s6d_samdec_img2tok_bf16(null, null, null, null, null, null, null, null, ctypes.c_float(1e-5), 1, 4090, 7, 128, 0, 0,
                                  null, null),
        L.s6d_samdec_tok2img_f32(null, null, 384, 300, 128, 0, null, 1, 4096, ctypes.c_float(0.25), null, null),
        L.s6d_crop_resize_pad_f32(null, null, null, 1, 480, 640, 0, f3, f3, null, null, null),
        L.s6d_sam_mask_post_f32(null, 1, 256, 1024, 1200, 1024, 480, 640, ctypes.c_float(0), ctypes.c_float(1), null, null, null),
        L.s6d_sam_mask_post_f32(null, 1, 256, 1024, 768, 1024, 480, 640, ctypes.c_float(0), ctypes.c_float(1), null, null, null),
        L.s6d_samdec_upscale_heads_bf16(null, null, null, ctypes.c_float(1e-6), null, null, null, 1, 5, 64, 64, 256, null, null),
    ]
    assert all(rc != 0 for rc in bad), bad
    with pytest.raises(RuntimeError):
        ops.sam_mask_post(torch.zeros(1, 3, 64, 32, device="cuda"), 256, (100, 100), (50, 50))      # not square
    with pytest.raises(RuntimeError):
        ops.samdec_tok2img(torch.zeros(2, 9, 128, device="cuda"), torch.zeros(2, 64, 256, device="cuda", dtype=torch.bfloat16),
                           0, 128, None, 0.25)



def test_round2_entry_points_reject_bad_arguments_and_accept_empty_batches(ops):
    """C-ABI argument checks of the entry points added in round 2 (GEMM, fused fine matching, coarse hypothesis kernels, the
    frame-batched ISM calls): empty batches return S6D_OK without touching a pointer, malformed shapes / NULL operands / misaligned
    rows are refused, nothing is launched; and the Python mirrors raise."""
    import ctypes

    from sam6d_amd import _lib
    L = _lib.lib()
    null, one = ctypes.c_void_p(0), ctypes.c_float(10.0)
    lg = ctypes.c_long
    L.s6d_fine_match_workspace_bytes.restype = ctypes.c_long
    ok = [
        L.s6d_gemm_bf16(null, lg(1280), null, lg(1280), null, null, lg(1280), 0, 1280, 1280, 0, 0, null),
        L.s6d_fine_match_f32(null, null, null, 0, 2049, 2049, 256, one, null, null, null, null, null),
        L.s6d_coarse_sample_f32(null, null, 0, 197, 197, 18000, null, null, null),
        L.s6d_smallest_k_f32(null, null, null, 0, 6000, 300, null, null, null, null),
        L.s6d_hypothesis_select_f32(null, null, null, null, 0, 300, 196, null, null, null),
        L.s6d_masked_depth_mean_frames_f32(null, null, null, 0, 480, 640, ctypes.c_float(1.0), null, null, null, null),
    ]
    assert all(rc == 0 for rc in ok), ok
    assert L.s6d_fine_match_workspace_bytes(32, 2049, 2049) > 0
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = ctypes.c_void_p(buf.data_ptr())
    bad = [
        L.s6d_gemm_bf16(p, lg(1280), p, lg(1280), null, p, lg(1280), 16, 1280, 1290, 0, 0, null),        # K % 64
        L.s6d_gemm_bf16(p, lg(1280), p, lg(1280), null, p, lg(1280), 16, 1000, 1280, 0, 0, null),        # N % 128
        L.s6d_gemm_bf16(p, lg(1276), p, lg(1280), null, p, lg(1280), 16, 1280, 1280, 0, 0, null),        # row stride not 16-byte
        L.s6d_gemm_bf16(p, lg(1280), p, lg(1280), null, p, lg(1280), 16, 1280, 1280, 2, 0, null),        # unknown epilogue
        L.s6d_gemm_bf16(null, lg(1280), p, lg(1280), null, p, lg(1280), 16, 1280, 1280, 0, 0, null),     # NULL operand, M > 0
        L.s6d_gemm_bf16_cblk(p, lg(1280), p, lg(1280), null, p, lg(1280), 16, 1280, 1280, 0, 60, 0, null),   # column block % 8
        L.s6d_fine_match_f32(p, p, p, 1, 1, 2049, 256, one, p, p, p, p, null),                           # no observed row
        L.s6d_fine_match_f32(p, p, p, 1, 2049, 2049, 128, one, p, p, p, p, null),                        # feature width
        L.s6d_fine_match_f32(p, p, p, 1, 2049, 2049, 256, ctypes.c_float(0.0), p, p, p, p, null),        # temperature
        L.s6d_fine_match_f32(p, p, p, 1, 2049, 2049, 256, one, null, p, p, p, null),                     # no workspace
        L.s6d_coarse_sample_f32(p, p, 1, 1, 197, 18000, p, p, null),
        L.s6d_coarse_sample_f32(null, p, 1, 197, 197, 18000, p, p, null),
        L.s6d_smallest_k_f32(p, p, p, 1, 100, 300, p, p, p, null),                                       # k > n
        L.s6d_hypothesis_select_f32(p, p, p, p, 1, 0, 196, p, p, null),
    ]
    assert all(rc != 0 for rc in bad), bad
    with pytest.raises(RuntimeError):
        ops.gemm_bf16(torch.zeros(4, 100, device="cuda", dtype=torch.bfloat16), torch.zeros(128, 100, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):
        ops.gemm_bf16(torch.zeros(4, 128, device="cuda"), torch.zeros(128, 128, device="cuda", dtype=torch.bfloat16))      # fp32 activations
    y = ops.gemm_bf16(torch.zeros(0, 128, device="cuda", dtype=torch.bfloat16), torch.zeros(256, 128, device="cuda", dtype=torch.bfloat16))
    assert y.shape == (0, 256)
