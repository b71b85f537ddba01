"""ISM scoring kernels executed on the HOST through the emulated HIP runtime: the per-kernel body of tests/test_gpu_ism.py
(cosine GEMM, top-k selection, split-bf16 MFMA patch scores on non-tile-aligned sizes) and the whole frame-scoring chain
against the reference golden."""
import os

import pytest

from tests import test_gpu_ism as T


def test_ism_kernels_on_the_emulator(emu):
    T.test_ism_kernels_individually_vs_oracle()


def test_frame_scoring_chain_on_the_emulator(emu):
    T.test_frame_scoring_vs_reference_golden("ism_scoring.npz")


def test_projection_bit_exact_on_the_emulator(emu):
    T.test_projection_is_bit_exact_given_the_reference_translation("ism_scoring.npz")
    T.test_projection_is_bit_exact_given_the_reference_translation("ism_scoring_p128.npz")


def test_masked_patch_similarity_methods_reach_the_kernel(emu):
    """MaskedPatch_MatrixSimilarity.compute_straight / compute_visible_ratio (detector.py:305,312 call
    them on a materialised (S,256,C) reference) go through patch_scores_kernel and agree with the oracle statements."""
    import torch

    from oracle import ism as oism
    from sam6d_amd.ism.loss import MaskedPatch_MatrixSimilarity
    from sam6d_amd.utils import synth
    d = synth.ism_inputs(P=5, O=2, T=3, C=64, n_patch=40, H=120, W=160, seed=9)
    q = d["qry_patch"]
    ref = d["ref_patch"][d["gt_obj"], d["gt_tem"]].contiguous()
    m = MaskedPatch_MatrixSimilarity(metric="cosine", chunk_size=64)
    calls = []
    orig = emu.patch_scores
    emu.patch_scores = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        appe = m.compute_straight(q, ref)
        ratio = m.compute_visible_ratio(q, ref, 0.5)
    finally:
        emu.patch_scores = orig
    assert len(calls) == 2
    assert torch.allclose(appe, oism.appearance_score(q, d['ref_patch'], d['gt_obj'], d['gt_tem'])[0], atol=1e-5)
    assert torch.allclose(ratio, oism.visible_ratio(q, ref, 0.5), atol=1e-5)


def test_half_descriptors_on_the_emulator(emu, monkeypatch):
    import torch
    T.test_half_descriptors_take_the_kernels(torch.bfloat16, monkeypatch)


def test_batched_frames_on_the_emulator(emu):
    T.test_batched_frames_equal_the_per_frame_loop()


def test_translation_sum_order_on_the_emulator(emu):
    """masked_depth_l1 / masked_depth_final follow ATen's CPU cascade order bit for bit at every remainder shape."""
    for H, W in ((120, 160), (37, 44), (130, 100), (2, 4), (96, 1028)):
        T.test_translation_sum_order_vs_oracle(H, W)
    T.test_frame_scoring_vs_reference_golden("ism_scoring_p128.npz")


def test_empty_selection_on_the_emulator(emu):
    T.test_score_frames_with_no_selected_proposal()


@pytest.mark.skipif(not os.environ.get("S6D_EMU_SLOW"), reason="2 minutes on the emulator; S6D_EMU_SLOW=1 runs it")
@pytest.mark.parametrize("name", ["ism_scoring_ycbv.npz", "ism_scoring_tless.npz"])
def test_large_configuration_goldens_on_the_emulator(emu, name):
    """BASELINE configs[2] / configs[3] sizes (YCB-V O=21, T-LESS P=256 / O=30): the kernels give the reference run's integer
    outputs bit for bit (last run on the host emulator: both green, 120 s)."""
    T.test_projection_is_bit_exact_given_the_reference_translation(name)
    T.test_frame_scoring_vs_reference_golden(name)
