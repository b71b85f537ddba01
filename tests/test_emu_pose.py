"""Pose-solver and point-transformer kernels executed on the HOST through the emulated HIP runtime: the bodies of
tests/test_gpu_pose.py at their small parametrisations (MFMA f32 / split-bf16 MFMA kernels included)."""
import pytest

from tests import test_gpu_pose as T


def test_rot_from_h_on_the_emulator(emu):
    T.test_rot_from_h_vs_svd(emu)


def test_weighted_procrustes_on_the_emulator(emu):
    T.test_weighted_procrustes_vs_oracle_and_batch_invariance(emu, 2, 197)
    T.test_weighted_procrustes_vs_oracle_and_batch_invariance(emu, 1, 7)


def test_half_stored_geo_embedding_on_the_emulator(emu):
    T.test_half_stored_geo_embedding_and_its_reader(emu)


def test_min_dist_on_the_emulator(emu):
    T.test_min_dist_vs_oracle(emu, 196, 3)


def test_rpe_attention_on_the_emulator(emu):
    T.test_rpe_attention_vs_oracle(emu, 3, 50, "1")
    T.test_rpe_attention_vs_oracle(emu, 3, 50, "0")


def test_geo_embedding_on_the_emulator(emu):
    T.test_geo_embedding_vs_oracle(emu, 1, 37)


@pytest.mark.parametrize("B,M", [(3, 300), (1, 17)])
def test_fine_assign_on_the_emulator(emu, B, M):
    T.test_fine_assign_vs_oracle(emu, B, M)


@pytest.mark.parametrize("B,M1,M2", [(1, 65, 65), (2, 40, 34), (1, 300, 270)])
def test_fine_match_on_the_emulator(emu, B, M1, M2):
    """Fused similarity + assignment (LDS-DMA staged split-bf16 tiles, three sweeps): single tile, ragged sides, several owner
    blocks / odd and even tile counts."""
    T._check_fine_match(emu, B, M1, M2)


def test_coarse_sampling_kernels_on_the_emulator(emu):
    T.test_coarse_sample_vs_oracle(emu, 2, 41, 900)
    T.test_coarse_sample_vs_oracle(emu, 2, 9, 300)          # small M1: the scratch is longer than the bins
    T.test_coarse_sample_vs_oracle(emu, 3, 21, 600)
    T.test_smallest_k_and_hypothesis_select_vs_library(emu)
    T.test_coarse_Rt_kernel_chain_vs_oracle(emu, 2, 40, 300, 30)


def test_positional_encoding_on_the_emulator(emu):
    """Same comparison as T.test_positional_encoding_fused_vs_oracle on a smaller cloud (the emulator is ~1e5 x slower than
    the GPU; the full-size body runs with S6D_EMU_SLOW=1)."""
    import os

    import torch

    from oracle import pem as opem
    from sam6d_amd.pem.pose_estimation_model import PositionalEncoding
    from sam6d_amd.utils import seeded, synth
    if os.environ.get("S6D_EMU_SLOW") == "1":
        return T.test_positional_encoding_fused_vs_oracle(emu)
    pe = seeded.load_seeded(PositionalEncoding(256).eval(), 6)
    W = {"PE." + k: v for k, v in pe.state_dict().items()}
    inp = synth.pem_inputs(1, seed=9, with_rgb=False)
    pts = inp["dense_po"][:, :160].contiguous()
    pts = pts / (pts.norm(dim=2).max(1)[0].reshape(-1, 1, 1) + 1e-6)
    with torch.no_grad():
        ref = opem.positional_encoding(W, "PE", pts)
        assert emu.have("pe_group")
        out = pe(pts)
    assert (out - ref).abs().max() < 5e-5, (out - ref).abs().max()


def test_pose_hypotheses_on_the_emulator(emu):
    T.test_pose_hypotheses_vs_oracle(emu)


def test_cross_and_linear_attention_on_the_emulator(emu):
    """The fused MHA-rows and focused-feature-map kernels inside the product layers vs the oracle layers, at reduced lengths
    (full-size body with S6D_EMU_SLOW=1)."""
    import os

    import torch

    from oracle import pem as opem
    from sam6d_amd.pem.layers import LinearTransformerLayer, TransformerLayer
    from sam6d_amd.utils import seeded
    if os.environ.get("S6D_EMU_SLOW") == "1":
        return T.test_cross_attention_and_linear_attention_vs_oracle(emu)
    g = torch.Generator().manual_seed(8)
    x, mem = torch.randn(2, 37, 256, generator=g), torch.randn(2, 50, 256, generator=g)
    tl = seeded.load_seeded(TransformerLayer(256).eval(), 5)
    W = {"t." + k: v for k, v in tl.state_dict().items()}
    with torch.no_grad():
        ref = opem.cross_layer(W, "t", x, mem)
        assert emu.have("mha") and emu.have("linear_attn_focus")
        out = tl(x, mem)
    assert (out - ref).abs().max() < 2e-5
    ll = seeded.load_seeded(LinearTransformerLayer(256).eval(), 6)
    W = {"l." + k: v for k, v in ll.state_dict().items()}
    xd, ms = torch.randn(1, 200, 256, generator=g), torch.randn(1, 60, 256, generator=g)
    with torch.no_grad():
        ref = opem.linear_layer(W, "l", xd, ms)
        out = ll(xd, ms)
    assert (out - ref).abs().max() < 2e-5
