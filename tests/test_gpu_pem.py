"""GPU parity of the drop-in PEM (sam6d_amd.pem) against the golden fixtures produced by the
reference modules and against the CPU oracle on the same seeded inputs.
Tolerances follow BASELINE.json: |R - R_ref|_F <= 1e-3, |t - t_ref| <= 1e-3 mm."""
import ast

import numpy as np
import pytest
import torch

from oracle import pem as opem
from sam6d_amd.utils import seeded, synth
from tests import util

pytestmark = pytest.mark.gpu

R_TOL = 1e-3          # Frobenius
T_TOL_M = 1e-6        # 1e-3 mm expressed in metres


@pytest.fixture(scope="module")
def net():
    assert torch.cuda.is_available()
    from sam6d_amd.pem import pose_estimation_model as pm
    n = pm.Net(pm.default_cfg()).eval()
    seeded.load_seeded(n, 1)
    return n.cuda()


def _to(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


def test_net_forward_vs_reference_golden(net):
    g = util.golden("pem_b2.npz")
    case = ast.literal_eval(str(g["case"]))
    inp = synth.pem_inputs(case["B"], seed=case["input_seed"])
    ep = {k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}
    ep["coarse_rand_u"] = synth.coarse_uniforms(case["B"], case["rand_seed"])
    with torch.no_grad():
        out = net(_to(ep, "cuda"))
    dR = np.linalg.norm(out["pred_R"].cpu().numpy() - g["net_pred_R"], axis=(1, 2))
    dt = np.abs(out["pred_t"].cpu().numpy() - g["net_pred_t"]).max()
    util.record_margin("net_forward_pem_b2_fp32vit", dR=dR.max(), dt_m=dt)
    # the fp32 feature path at the stated bar (the case is chaotic beyond fp32-class feature noise:
    # tests/test_host_pem.py::test_conditioning_of_the_two_net_forward_cases)
    assert dR.max() <= R_TOL and dt <= T_TOL_M, (dR, dt)
    np.testing.assert_allclose(out["pred_pose_score"].cpu().numpy(), g["net_pred_pose_score"], atol=5e-3)


@pytest.mark.parametrize("vit", ["fp32", "fp16", "bf16"])
def test_net_forward_well_conditioned_vs_reference_golden(net, vit, monkeypatch):
    """Net.forward on the well-conditioned frame of tests/golden/pem_wc.npz (template features = the reference feature
    extractor's own output for the observed pixels; pose by the reference Net).  fp32 ViT-B and the fused IEEE-half ViT-B
    (fp16: what bench.py runs since round 3): |R - R_ref|_F <= 1e-3 and |t - t_ref| <= 1e-3 mm, north_star's bar.  bf16 ViT-B
    (kept as an option): rotation at the same bar; its translation carries the linear response of the matcher to bf16-class
    feature noise measured on the oracle (|dt| ~ 3e-4 m per unit relative noise; features 7.6e-3 off): bound 1e-2 mm, measured
    1.3e-3 mm -- which is why it is not the benched dtype.  The half-precision runs must go through the fused pipeline."""
    g = util.golden("pem_wc.npz")
    case = ast.literal_eval(str(g["case"]))
    inp = synth.pem_inputs(case["B"], seed=case["input_seed"])
    ep = {k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}
    W = util.pem_weights(case["weight_seed"])
    with torch.no_grad():
        ep["dense_fo"] = opem.feature_extraction(W, ep)[1]          # the checker's fp32 features stand for the template store
    util.assert_digest_close(ep["dense_fo"], g["fo_sum"], g["fo_smp"], 4099, 1e-4, 1e-5, "template features")
    ep["coarse_rand_u"] = synth.coarse_uniforms(case["B"], case["rand_seed"])
    monkeypatch.setenv("S6D_PEM_VIT_DTYPE", vit)
    from sam6d_amd import ops
    calls = []
    real = ops.seq_attention
    monkeypatch.setattr(ops, "seq_attention", lambda *a, **k: (calls.append(a[0].dtype), real(*a, **k))[1])
    with torch.no_grad():
        out = net(_to(ep, "cuda"))
    if vit != "fp32":                                             # 12 blocks through the fused attention kernel of that element type
        assert calls == [dict(fp16=torch.float16, bf16=torch.bfloat16)[vit]] * 12, calls
    dR = np.linalg.norm(out["pred_R"].cpu().numpy() - g["net_pred_R"], axis=(1, 2)).max()
    dt = np.abs(out["pred_t"].cpu().numpy() - g["net_pred_t"]).max()
    util.record_margin(f"net_forward_pem_wc_{vit}vit", dR=dR, dt_m=dt)
    assert dR <= R_TOL and dt <= (10 * T_TOL_M if vit == "bf16" else T_TOL_M), (vit, dR, dt)


def test_known_answer_vs_reference_golden_and_truth(net):
    g = util.golden("pem_b2.npz")
    case = ast.literal_eval(str(g["case"]))
    inp = synth.pem_inputs(case["B"], seed=case["input_seed"], with_rgb=False)
    radius = torch.norm(inp["dense_po"], dim=2).max(1)[0]
    s = radius.reshape(-1, 1, 1) + 1e-6
    ep = _to(dict(model=inp["model"], coarse_rand_u=synth.coarse_uniforms(case["B"], case["rand_seed"])), "cuda")
    with torch.no_grad():
        out = net.match((inp["pts"] / s).cuda(), inp["dense_fm_kat"].cuda(), (inp["dense_po"] / s).cuda(),
                        inp["dense_fo"].cuda(), radius.cuda(), ep)
    dR = np.linalg.norm(out["pred_R"].cpu().numpy() - g["kat_pred_R"], axis=(1, 2))
    dt = np.abs(out["pred_t"].cpu().numpy() - g["kat_pred_t"]).max()
    assert dR.max() <= R_TOL and dt <= T_TOL_M, (dR, dt)
    assert np.linalg.norm(out["pred_R"].cpu().numpy() - g["kat_gt_R"], axis=(1, 2)).max() < 1e-3
    np.testing.assert_allclose(out["pred_pose_score"].cpu().numpy(), g["kat_pred_pose_score"], atol=2e-3)


def test_known_answer_at_the_benched_batch_vs_reference_golden(net):
    """B = 32 (BASELINE configs[1], the batch bench.py runs): the matching path against tests/golden/pem_b32.npz, produced by
    the reference's own sub-modules (oracle/gen_golden.py pem_b32).  |R - R_ref|_F <= 1e-3, |t - t_ref| <= 1e-3 mm, identical ADD recall."""
    from sam6d_amd.utils import metrics
    g = util.golden("pem_b32.npz")
    case = ast.literal_eval(str(g["case"]))
    B = case["B"]
    inp = synth.pem_inputs(B, seed=case["input_seed"], with_rgb=False)
    radius = torch.norm(inp["dense_po"], dim=2).max(1)[0]
    s = radius.reshape(-1, 1, 1) + 1e-6
    ep = _to(dict(model=inp["model"], coarse_rand_u=synth.coarse_uniforms(B, case["rand_seed"])), "cuda")
    with torch.no_grad():
        out = net.match((inp["pts"] / s).cuda(), inp["dense_fm_kat"].cuda(), (inp["dense_po"] / s).cuda(),
                        inp["dense_fo"].cuda(), radius.cuda(), ep)
    R, t = out["pred_R"].cpu(), out["pred_t"].cpu()
    dR = np.linalg.norm(R.numpy() - g["kat_pred_R"], axis=(1, 2))
    dt = np.abs(t.numpy() - g["kat_pred_t"]).max()
    assert dR.max() <= R_TOL and dt <= T_TOL_M, (dR.max(), dt)
    np.testing.assert_allclose(out["pred_pose_score"].cpu().numpy(), g["kat_pred_pose_score"], atol=2e-3)
    a = metrics.add_recall(R, t, inp["gt_R"], inp["gt_t"], inp["model"], 0.2)
    b = metrics.add_recall(torch.from_numpy(g["kat_pred_R"]), torch.from_numpy(g["kat_pred_t"]), inp["gt_R"], inp["gt_t"], inp["model"], 0.2)
    assert torch.equal(a[1], b[1]) and a[0] == b[0]


@pytest.mark.parametrize("B,seed", [(4, 21), (16, 33)])
def test_known_answer_vs_oracle_other_seeds(net, B, seed):
    """Same seeded inputs through the CPU oracle and the MI355X path."""
    W = util.pem_weights(1)
    inp = synth.pem_inputs(B, seed=seed, with_rgb=False)
    radius = torch.norm(inp["dense_po"], dim=2).max(1)[0]
    s = radius.reshape(-1, 1, 1) + 1e-6
    ru = synth.coarse_uniforms(B, seed + 1)
    with torch.no_grad():
        ref = opem.matching_forward(W, inp["pts"] / s, inp["dense_fm_kat"], inp["dense_po"] / s, inp["dense_fo"],
                                    radius, inp["model"], ru)
        ep = _to(dict(model=inp["model"], coarse_rand_u=ru), "cuda")
        out = net.match((inp["pts"] / s).cuda(), inp["dense_fm_kat"].cuda(), (inp["dense_po"] / s).cuda(),
                        inp["dense_fo"].cuda(), radius.cuda(), ep)
    dR = (out["pred_R"].cpu() - ref["pred_R"]).norm(dim=(1, 2))
    dt = (out["pred_t"].cpu() - ref["pred_t"]).abs().max()
    assert dR.max() <= R_TOL and dt <= T_TOL_M, (dR, dt)
    # ADD(-S) style recall identical on identical inputs
    from sam6d_amd.utils import metrics
    m = inp["model"]
    a = metrics.add_recall(out["pred_R"].cpu(), out["pred_t"].cpu(), inp["gt_R"], inp["gt_t"], m, 0.2)
    b = metrics.add_recall(ref["pred_R"], ref["pred_t"], inp["gt_R"], inp["gt_t"], m, 0.2)
    assert torch.equal(a[1], b[1]) and a[0] == b[0]


def test_bf16_vit_features_keep_the_pose(net, monkeypatch):
    """bench.py runs the PEM ViT-B in bf16 (BASELINE config: bf16).  With template features taken from the same
    (fp32) feature extractor, the matcher must recover the synthetic pose with either ViT precision, and the two
    runs must agree within the parity tolerance."""
    B = 4
    inp = synth.pem_inputs(B, seed=77)
    ep = {k: inp[k].cuda() for k in ("pts", "rgb", "rgb_choose", "model", "dense_po")}
    with torch.no_grad():
        monkeypatch.setenv("S6D_PEM_VIT_DTYPE", "fp32")
        ep["dense_fo"] = net.feature_extraction.get_img_feats(ep["rgb"], ep["rgb_choose"])
        ru = synth.coarse_uniforms(B, 78).cuda()
        out32 = net(dict(ep, coarse_rand_u=ru))
        monkeypatch.setenv("S6D_PEM_VIT_DTYPE", "bf16")
        out16 = net(dict(ep, coarse_rand_u=ru))
    gt = inp["gt_R"]
    e32 = (out32["pred_R"].cpu() - gt).norm(dim=(1, 2))
    e16 = (out16["pred_R"].cpu() - gt).norm(dim=(1, 2))
    d = (out16["pred_R"] - out32["pred_R"]).norm(dim=(1, 2)).cpu()
    dt = (out16["pred_t"] - out32["pred_t"]).abs().max().item()
    util.record_margin("bf16_vs_fp32_vit_features", e32=e32.max(), e16=e16.max(), dR=d.max(), dt_m=dt)
    # both recover the synthetic pose (truth carries the 1e-4 m point noise) and agree with each other at the rotation bar;
    # translation: the matcher's linear response to bf16-class feature noise (see test_net_forward_well_conditioned_*)
    assert e32.max() < 2e-3 and e16.max() < 2e-3 and d.max() <= R_TOL, (e32, e16, d)
    assert dt <= 10 * T_TOL_M, dt


def test_upsample_gather_kernel_vs_dense_reference(net):
    """Chosen-pixel features without the 224x224 map (fused kernel) == reference-shaped dense map + gather."""
    from sam6d_amd import ops
    assert ops.have("upsample_gather")
    ae = net.feature_extraction.rgb_net
    inp = synth.pem_inputs(2, seed=3)
    g = torch.Generator().manual_seed(4)
    choose = torch.randint(0, 224 * 224, (2, 2048), generator=g)
    choose[:, :6] = torch.tensor([0, 223, 224 * 223, 224 * 224 - 1, 3, 224 * 3])      # borders / clamps
    with torch.no_grad():
        fm, _ = ae(inp["rgb"].cuda())
        exp = torch.gather(fm.flatten(2), 2, choose.cuda().unsqueeze(1).expand(-1, 256, -1)).transpose(1, 2)
        got = ae.sample(inp["rgb"].cuda(), choose.cuda())
    assert torch.allclose(got, exp, atol=2e-5, rtol=1e-5), (got - exp).abs().max()


def test_real_example_frame_through_preprocessing_and_net(net):
    """BASELINE configs[0]: the reference's Data/Example frame (tests/golden/example_frame.npz) through the device pre-processing
    (kernel path) and Net.forward: sampled points and pixel indices identical to the oracle, pose within the parity tolerance of
    the reference Net's on the same tensors."""
    from sam6d_amd.pem import preprocess as pre
    from tests.test_host_example_frame import frame
    g = util.golden("example_frame.npz")
    depth, mask, keys, case = frame(g)
    out = pre.observed_inputs(torch.from_numpy(g["rgb"]).cuda(), torch.from_numpy(depth).cuda(), torch.from_numpy(g["K"]),
                              torch.from_numpy(mask[None]).cuda(), float(g["radius"]), keys=keys.cuda())
    assert out["kept"].cpu().tolist() == g["oracle_kept"].tolist()
    np.testing.assert_array_equal(out["pts"].cpu().numpy(), g["oracle_pts"])
    np.testing.assert_array_equal(out["rgb_choose"].cpu().numpy(), g["oracle_rgb_choose"])
    dense_fo = torch.randn(1, 2048, 256, generator=torch.Generator().manual_seed(case["feat_seed"]))
    ep = dict(pts=out["pts"], rgb=out["rgb"], rgb_choose=out["rgb_choose"], model=torch.from_numpy(g["model"])[None].cuda(),
              dense_po=torch.from_numpy(g["dense_po"])[None].cuda(), dense_fo=dense_fo.cuda(),
              coarse_rand_u=synth.coarse_uniforms(1, case["rand_seed"]).cuda())
    import os
    old = os.environ.get("S6D_PEM_VIT_DTYPE")
    os.environ["S6D_PEM_VIT_DTYPE"] = "fp32"; __import__("sam6d_amd.policy").policy.reload()                        # parity run: the fp32 ViT (bf16 is the throughput configuration)
    try:
        with torch.no_grad():
            res = net(ep)
    finally:
        if old is None:
            os.environ.pop("S6D_PEM_VIT_DTYPE"); __import__("sam6d_amd.policy").policy.reload()
        else:
            os.environ["S6D_PEM_VIT_DTYPE"] = old; __import__("sam6d_amd.policy").policy.reload()
    dR = np.linalg.norm(res["pred_R"].cpu().numpy() - g["ref_pred_R"], axis=(1, 2))
    dt = np.abs(res["pred_t"].cpu().numpy() - g["ref_pred_t"]).max()
    util.record_margin("example_frame_fp32vit", dR=dR.max(), dt_m=dt)
    assert dR.max() <= R_TOL and dt <= T_TOL_M, (dR, dt)


def test_nonfinite_rows_kernel_flags_exactly_the_rows_with_inf_or_nan():
    """s6d_nonfinite_rows_f32 (the range guard's one read of the up-projection): rows with an inf or a NaN anywhere -- first element,
    last element, a single one in 800 k -- are flagged and no other; large FINITE values (whose library SUM would overflow to inf,
    a false alarm of round 4's first form) are not."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 196, 64, generator=g)
    x[1, 0, 0] = float("inf")
    x[2, -1, -1] = float("nan")
    x[3, 100, 7] = float("-inf")
    x[4] = 3.0e38                                      # finite; the sum of the row is inf in fp32
    out = ops.nonfinite_rows(x.cuda()).cpu()
    assert out.tolist() == [False, True, True, True, False, False]
    assert ops.nonfinite_rows(torch.zeros(0, 8).cuda()).shape == (0,)
    big = torch.zeros(2, 802816)
    big[1, 777777] = float("nan")
    assert ops.nonfinite_rows(big.cuda()).cpu().tolist() == [False, True]


def test_fp16_extractor_range_guard_reruns_overflowing_instances_in_fp32(net, monkeypatch):
    """VERDICT r3 missing #7: the IEEE-half ViT-B has the range 65504; a released checkpoint with an outlier channel must not turn
    the pose into NaN silently.  An outlier of 1e5 is planted in one channel of the residual stream (fc2 bias of block 3): every
    instance overflows in half, the guarded forward flags them, warns and returns the fp32 extractor's poses bit for bit; with the
    guard's host read off the flag is still returned and the poses are not the fp32 ones."""
    import warnings

    from sam6d_amd.pem import pose_estimation_model as pm
    B = 3
    inp = synth.pem_inputs(B, seed=5)
    ep = _to({k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}, "cuda")
    ep["coarse_rand_u"] = synth.coarse_uniforms(B, 6).cuda()
    bad_net = pm.Net(pm.default_cfg()).eval().cuda()
    bad_net.load_state_dict(net.state_dict())
    with torch.no_grad():
        bad_net.feature_extraction.rgb_net.vit.blocks[3].mlp.fc2.bias[5] = 1.0e5
        monkeypatch.setenv("S6D_PEM_VIT_DTYPE", "fp32")
        ref = bad_net(dict(ep))
        assert "f16_overflow" not in ref and torch.isfinite(ref["pred_R"]).all()
        monkeypatch.setenv("S6D_PEM_VIT_DTYPE", "fp16")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = bad_net(dict(ep))
        assert any("overflowed" in str(x.message) for x in w), [str(x.message) for x in w]
        assert out["f16_overflow"].all()
        assert torch.equal(out["pred_R"], ref["pred_R"]) and torch.equal(out["pred_t"], ref["pred_t"])
        monkeypatch.setenv("S6D_PEM_F16_GUARD", "0")
        raw = bad_net(dict(ep))
        assert raw["f16_overflow"].all() and not torch.equal(raw["pred_R"], ref["pred_R"])
        monkeypatch.delenv("S6D_PEM_F16_GUARD")
        # the healthy network: no flag set, no warning, nothing re-run
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ok = net(dict(ep))
        assert not ok["f16_overflow"].any() and not any("overflowed" in str(x.message) for x in w)


def test_range_guard_flag_survives_graph_capture_and_replay(net, monkeypatch):
    """FramePipeline._pem_forward captures Net.forward (IEEE-half extractor) into a hipGraph and reads the guard's flag after every
    replay.  The flag kernel must work inside a capture: a healthy network replays with no flag set and no fp32 re-run (round 4: a
    memset node in the capture left the flags unset, every replay was followed by an eager fp32 forward -- 31 instead of 16 ms per
    10 instances), and an overflowing one is still caught."""
    import warnings

    from sam6d_amd import pipeline
    from sam6d_amd.pem import pose_estimation_model as pm
    monkeypatch.setenv("S6D_PEM_VIT_DTYPE", "fp16")
    B = 4
    inp = synth.pem_inputs(B, seed=7)
    ep = _to({k: inp[k] for k in ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo")}, "cuda")
    ep["coarse_rand_u"] = synth.coarse_uniforms(B, 6).cuda()
    for bad in (False, True):
        n = pm.Net(pm.default_cfg()).eval().cuda()
        n.load_state_dict(net.state_dict())
        if bad:
            with torch.no_grad():
                n.feature_extraction.rgb_net.vit.blocks[3].mlp.fc2.bias[5] = 1.0e5
        calls = []

        class Counted:                                                  # the pipeline's own calls of the network (not the guard's inner re-run)
            def __call__(self, e):
                calls.append(1)
                return n(e)

            def __getattr__(self, name):
                return getattr(n, name)
        fp = pipeline.FramePipeline.__new__(pipeline.FramePipeline)
        fp.pem, fp.graph_max, fp._pem_graphs = Counted(), 16, {}
        with torch.no_grad(), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            fp._pem_forward(dict(ep))                                   # warm-up x 2 + capture (+ the re-run when flagged)
            k = len(calls)
            out = fp._pem_forward(dict(ep))                             # replay
        reran = len(calls) - k
        assert reran == (1 if bad else 0), (bad, reran)
        assert any("overflowed" in str(x.message) for x in w) == bad
        assert torch.isfinite(out["pred_R"]).all()
