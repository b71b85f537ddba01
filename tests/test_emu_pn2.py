"""PointNet++ kernels (FPS, ball query, gathers) executed on the HOST through the emulated HIP runtime: the bodies of
tests/test_gpu_pn2.py at small sizes -- bit-exact against the C oracle, tie-break rule included."""
import pytest

from tests import test_gpu_pn2 as T


@pytest.mark.parametrize("B,N,M,dup", [(2, 64, 64, False), (1, 1, 1, False), (1, 700, 64, True), (2, 300, 20, True), (1, 2048, 40, False)])
def test_fps_on_the_emulator(emu, B, N, M, dup):
    T.test_fps_bit_exact(emu, B, N, M, dup)


def test_fps_tie_break_on_the_emulator(emu):
    T.test_fps_tie_break_rule(emu)


@pytest.mark.parametrize("B,N,M,r,ns", [(1, 300, 50, 0.2, 16), (1, 100, 7, 0.5, 128), (2, 500, 64, 0.1, 32)])
def test_ball_query_on_the_emulator(emu, B, N, M, r, ns):
    T.test_ball_query_bit_exact(emu, B, N, M, r, ns)


def test_gathers_on_the_emulator(emu):
    T.test_ball_query_no_neighbour_gives_zeros(emu)
    T.test_gather_group_exact(emu)


def test_edge_cases_on_the_emulator(emu):
    """Bodies of tests/test_gpu_edge_cases.py: M == N permutation, single point, nsample larger than the cloud with coincident
    points (first-hit fill)."""
    from tests import test_gpu_edge_cases as E
    E.test_fps_all_points_and_single_point(emu)
    E.test_ball_query_nsample_larger_than_cloud_and_coincident_points(emu)
