"""s6d_linear_f32 on the emulator: the bodies of tests/test_gpu_plin.py."""
from tests import test_gpu_plin as T


def test_linear_f32_on_the_emulator(emu):
    for case in T.CASES:
        T.test_linear_f32_vs_library(*case)
