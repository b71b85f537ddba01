"""TEST INFRASTRUCTURE (CPU oracle) -- the summation ORDER of ATen's CPU ``sum`` over a contiguous inner dimension.

The ISM's query translation (reference ``utils/trimesh_utils.py:77-105``) is three ``torch.sum(.., axis=(1, 2))`` calls over
(S, H, W) maps: Z in float32, X and Y in float64.  On the CPU, ATen reduces each map with ``cascade_sum`` /
``vectorized_inner_sum`` (aten/src/ATen/native/cpu/SumKernel.cpp of the pinned torch 2.10): the flat map is read as rows of
4 * V elements (V = lanes of the AVX2 vector the kernel is compiled for: 8 floats, 4 doubles -- also on AVX-512 hosts, where
this kernel is not re-specialised; checked against ``torch.sum`` in tests/test_oracle_golden.py), every COLUMN is summed on its
own through a 4-level cascade with 2**p rows per level step (p = max(4, ceil_log2(rows) // 4)), the four vectors of a row are
then added k = 1, 2, 3 onto vector 0, and the V lanes are added in lane order onto the scalar tail.  Restated here in numpy so
that the order is pinned independently of the host that runs the tests; the device kernel ``masked_depth_*`` in
``csrc/s6d_ism.hip`` follows the same tree.
"""
import numpy as np


def ceil_log2(x):
    """c10::utils::CeilLog2."""
    return 1 if x <= 2 else int(x - 1).bit_length()


def cascade_sum_rows(x, V):
    """x (S, n) float32/float64 -> (S,) sums in ATen's CPU order for a contiguous inner reduction of n >= V elements."""
    x = np.ascontiguousarray(x)
    S, n = x.shape
    dt = x.dtype
    if n < V:
        raise NotImplementedError("ATen takes a scalar path below one vector")
    vec_size = n // V
    rows_n = vec_size // 4
    C = 4 * V
    rows = x[:, : rows_n * C].reshape(S, rows_n, C)
    p = max(4, ceil_log2(rows_n) // 4)
    step, mask = 1 << p, (1 << p) - 1
    acc = np.zeros((4, S, C), dt)
    i = 0
    while i + step <= rows_n:
        for _ in range(step):
            acc[0] += rows[:, i]
            i += 1
        for j in range(1, 4):
            acc[j] += acc[j - 1]
            acc[j - 1] = 0
            if i & (mask << (j * p)):
                break
    while i < rows_n:
        acc[0] += rows[:, i]
        i += 1
    for j in range(1, 4):
        acc[0] += acc[j]
    ps = acc[0].reshape(S, 4, V).copy()
    for r in range(rows_n * 4, vec_size):
        ps[:, 0] += x[:, r * V:(r + 1) * V]
    for k in range(1, 4):
        ps[:, 0] += ps[:, k]
    fin = np.zeros(S, dt)
    for k in range(vec_size * V, n):
        fin = fin + x[:, k]
    for k in range(V):
        fin = fin + ps[:, 0, k]
    return fin


def sum_f32(x):
    return cascade_sum_rows(np.asarray(x, np.float32), 8)


def sum_f64(x):
    return cascade_sum_rows(np.asarray(x, np.float64), 4)
