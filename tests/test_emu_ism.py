"""ISM scoring kernels executed on the HOST through the emulated HIP runtime: the per-kernel body of tests/test_gpu_ism.py
(cosine GEMM, top-k selection, split-bf16 MFMA patch scores on non-tile-aligned sizes) and the whole frame-scoring chain
against the reference golden."""
from tests import test_gpu_ism as T


def test_ism_kernels_on_the_emulator(emu):
    T.test_ism_kernels_individually_vs_oracle()


def test_frame_scoring_chain_on_the_emulator(emu):
    T.test_frame_scoring_vs_reference_golden("ism_scoring.npz")


def test_projection_bit_exact_on_the_emulator(emu):
    T.test_projection_is_bit_exact_given_the_reference_translation("ism_scoring.npz")
    T.test_projection_is_bit_exact_given_the_reference_translation("ism_scoring_p128.npz")
