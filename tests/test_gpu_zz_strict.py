"""Strict mode (sam6d_amd/policy.py): every module is "if the kernel applies: kernel, else the reference's torch
statements"; a guard that stops matching would drop to rocBLAS / ATen on the GPU with no signal, and the parity tests would still
pass (the library branch IS the reference's arithmetic).  Under `strict` such a branch raises and names the failed guard; without
it the branch is counted.  Here: the benched step (bench.py::HotPath = BASELINE configs[1]'s three stages at the benched policy)
and the ISM scoring sizes of configs[2] / [3] run with ZERO library branches, and a guard that fails is reported by name."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_benched_step_takes_no_library_branch():
    import bench
    from sam6d_amd import policy
    dev = torch.device("cuda", 0)
    with policy.use(strict="1", pem_vit_dtype="fp16"):
        hp = bench.HotPath(dev, 2, 2)                      # two frames: every stage at its benched kernels, small batch
        policy.reset_library_branch_hits()
        rec = hp.step()
        torch.cuda.synchronize()
        assert torch.isfinite(rec).all()
        assert policy.library_branch_hits() == {}


@pytest.mark.parametrize("P,O,T", [(128, 1, 42), (64, 21, 42), (256, 30, 42)])   # BASELINE configs[1], [2] (YCB-V), [3] (T-LESS)
def test_ism_scoring_sizes_take_no_library_branch(P, O, T):
    from sam6d_amd import policy
    from sam6d_amd.ism.scoring import FrameScorer
    from sam6d_amd.utils import synth
    d = synth.ism_inputs(P=P, O=O, T=T, seed=5)
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    with policy.use(strict="1"):
        policy.reset_library_branch_hits()
        sc = FrameScorer(d["ref_cls"], d["ref_patch"], d["poses"], d["pointcloud"], confidence_thresh=0.2)
        out = sc.score_frames(d["qry_cls"][None], d["qry_patch"][None], d["masks"][None], d["boxes"][None], d["depth"][None], d["K"][None])
        torch.cuda.synchronize()
        assert out["final"].numel() > 0
        assert policy.library_branch_hits() == {}


def test_a_failed_guard_is_named():
    """The PEM's ViT-B extractor in float32 is the library GEMM path (the reference's precision, the library default): strict mode
    says so, by site and guard; without strict mode the branch is counted."""
    from sam6d_amd import policy
    from sam6d_amd.pem import pose_estimation_model as pm
    from sam6d_amd.utils import seeded, synth
    net = seeded.load_seeded(pm.Net(pm.default_cfg()).eval(), 1).cuda()
    ep = {k: v.cuda() for k, v in synth.pem_inputs(B=1, seed=3).items()}
    with torch.no_grad():
        with policy.use(strict="1", pem_vit_dtype="fp32"):
            with pytest.raises(policy.StrictError, match=r"(utils\.fused_linear|pem\.ViT\.forward).*guard `half_dtype` failed"):
                net(dict(ep))
        with policy.use(pem_vit_dtype="fp32"):
            policy.reset_library_branch_hits()
            net(dict(ep))
            hits = policy.library_branch_hits()
            assert ("pem.ViT.forward", "half_dtype") in hits and ("utils.fused_linear", "half_dtype") in hits, hits
        with policy.use(strict="1", pem_vit_dtype="fp16"):
            policy.reset_library_branch_hits()
            out = net(dict(ep))
            assert torch.isfinite(out["pred_R"]).all() and policy.library_branch_hits() == {}
