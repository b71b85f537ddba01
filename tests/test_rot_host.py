"""The closed-form 3x3 Procrustes rotation used by the pose kernels (csrc/s6d_rot.h) is plain
C++: compile it for the host and compare with the SVD formula of the reference
(model_utils.py:343-347) -- a logic check that needs no GPU."""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib(tmp_path):
    so = str(tmp_path / "librot_host.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "host_cc", "rot_host.cc")])
    L = ctypes.CDLL(so)
    L.rot_from_h_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return L


def _ref(H):
    U, _, V = torch.svd(H)
    eye = torch.eye(3, dtype=H.dtype).repeat(H.shape[0], 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ U.transpose(1, 2)))
    return V @ eye @ U.transpose(1, 2)


def test_rot_from_h_matches_svd_formula(tmp_path):
    L = _lib(tmp_path)
    g = torch.Generator().manual_seed(0)
    full = torch.randn(2000, 3, 3, generator=g, dtype=torch.float64)
    # rank-2 matrices as produced by 3-point hypotheses (centred triangles)
    a = torch.randn(2000, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(2000, 3, 3, generator=g, dtype=torch.float64)
    a = a - a.mean(1, keepdim=True)
    b = b - b.mean(1, keepdim=True)
    tri = a.transpose(1, 2) @ b
    refl = full.clone()
    refl[:, :, 0] *= -1
    for H in (full, tri, refl, 1e-6 * tri, 1e3 * full):
        Hc = np.ascontiguousarray(H.numpy())
        R = np.zeros_like(Hc)
        L.rot_from_h_host(Hc.ctypes.data, H.shape[0], R.ctypes.data)
        ref = _ref(H).numpy()
        err = np.abs(R - ref).reshape(len(R), -1).max(1)
        # ill-conditioned cases (two nearly equal / nearly zero singular values) are arbitrary in both
        s = torch.linalg.svdvals(H).numpy()
        well = (s[:, 1] - s[:, 2] > 1e-3 * s[:, 0]) & (s[:, 0] - s[:, 1] > 1e-6 * s[:, 0])
        assert err[well].max() < 1e-8, err[well].max()
        Rt = torch.from_numpy(R)
        assert torch.allclose(Rt @ Rt.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand_as(Rt), atol=1e-9)
        assert torch.allclose(torch.det(Rt), torch.ones(len(R), dtype=torch.float64), atol=1e-9)


def test_rot_from_h_zero_matrix_is_identity(tmp_path):
    L = _lib(tmp_path)
    H = np.zeros((1, 3, 3))
    R = np.ones((1, 3, 3))
    L.rot_from_h_host(H.ctypes.data, 1, R.ctypes.data)
    assert np.array_equal(R[0], np.eye(3))
