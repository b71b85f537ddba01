"""Row-order segment sums (csrc/s6d_seqsum.h, the kernel behind s6d_segment_seq_sum_f32): the kernel source compiled for
the host with the wave emulated by 64 threads (tests/host_cc/seqsum_host.cc) must equal numpy's add.reduce over axis 0 bit for
bit -- and with that centroid the PEM pre-processing agrees with the oracle at radii where the float64-accumulated centroid
does not."""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib(tmp_path):
    so = str(tmp_path / "libseqsum_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                           os.path.join(HERE, "host_cc", "seqsum_host.cc")])
    L = ctypes.CDLL(so)
    L.segment_seq_sum_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return L


def _run(L, x, start, count):
    out = np.zeros((len(start), x.shape[1]), dtype=np.float32)
    x, start, count = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(start, np.int64), np.ascontiguousarray(count, np.int64)
    L.segment_seq_sum_host(x.ctypes.data, start.ctypes.data, count.ctypes.data, len(start), x.shape[1], out.ctypes.data)
    return out


def test_kernel_source_equals_numpy_row_order(tmp_path):
    L = _lib(tmp_path)
    rng = np.random.default_rng(0)
    for C in (2, 3, 4):
        counts = np.array([0, 1, 2, 7, 8, 9, 511, 512, 513, 1024, 1025, 3000, 20001])
        x = (rng.standard_normal((counts.sum(), C)) * 0.3 + 0.8).astype(np.float32)
        start = np.cumsum(counts) - counts
        got = _run(L, x, start, counts)
        for i, (s0, c) in enumerate(zip(start, counts)):
            want = np.add.reduce(x[s0:s0 + c], axis=0) if c else np.zeros(C, np.float32)
            np.testing.assert_array_equal(got[i], want, err_msg=f"C={C} n={c}")
    # the order matters: a float64 sum of the same rows is a different float32 number for long segments, and so is numpy's own
    # sum of ONE column (C = 1 makes the reduced axis the contiguous one, where numpy switches to pairwise summation -- the
    # reason the entry point refuses C = 1)
    long = x[start[-1]:]
    assert not np.array_equal(np.add.reduce(long, axis=0), long.astype(np.float64).sum(0).astype(np.float32))
    col = np.ascontiguousarray(long[:, :1])
    assert not np.array_equal(_run(L, col, np.array([0]), np.array([len(col)]))[0], np.add.reduce(col, axis=0))


def test_preprocessing_with_the_sequential_centroid_matches_the_oracle(tmp_path, monkeypatch):
    """observed_inputs == oracle loop at the radii where a float64-accumulated centroid (round 1's default, deleted in round 2)
    flipped boundary points: with the module's own sequential float32 centroid, and with the centroid kernel's arithmetic
    (through the host build) in its place."""
    from oracle import pem_pre as opre
    from sam6d_amd.pem import preprocess as pre
    from sam6d_amd.utils import synth
    L = _lib(tmp_path)
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    radius = np.array([0.12, 0.03, 0.5, 0.12, 0.06, 0.2, 0.07, 0.01])
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), radius,
                                keys=inp["keys"].numpy(), **kw)
    args = (torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], torch.from_numpy(radius))
    default = pre.observed_inputs(*args, keys=inp["keys"], **kw)
    np.testing.assert_array_equal(default["pts"].numpy(), ref["pts"])         # round 1's float64 centroid differed here

    def seq_sum(x, start, count):
        return torch.from_numpy(_run(L, x.numpy(), start.numpy(), count.numpy()))
    monkeypatch.setattr(pre, "_segment_seq_sum", seq_sum)
    out = pre.observed_inputs(*args, keys=inp["keys"], **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])
    np.testing.assert_array_equal(out["rgb"].numpy(), ref["rgb"])
