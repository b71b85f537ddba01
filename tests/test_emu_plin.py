"""s6d_linear_f32 on the emulator: the bodies of tests/test_gpu_plin.py."""
from tests import test_gpu_plin as T


def test_linear_f32_on_the_emulator(emu):
    for case in T.CASES:
        T.test_linear_f32_vs_library(*case)


def test_linear_attention_on_the_emulator(emu):
    T.test_linear_attention_vs_the_library_statement(2, 100, 37)
    T.test_linear_attention_vs_the_library_statement(3, 64, 196)


def test_attention_output_chain_on_the_emulator(emu):
    T.test_attention_output_chain_equals_the_three_launches_bit_for_bit("mha", 1, 45)
    T.test_attention_output_chain_equals_the_three_launches_bit_for_bit("linear", 1, 70)
