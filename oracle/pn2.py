"""ctypes front-end of ``pn2_oracle.c`` (test infrastructure only).

Signatures mirror the reference pybind module ``pointnet2._ext``
(reference: Pose_Estimation_Model/model/pointnet2/_ext_src/src/bindings.cpp:11-24)
but run on CPU torch tensors, so the reference's own PEM modules can be
executed in a container without a GPU (see ``oracle/gen_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libs6d_oracle.so")
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
        ci, cf = ctypes.c_int, ctypes.c_float
        L.s6d_oracle_fps.argtypes = [fp, ci, ci, ci, ip]
        L.s6d_oracle_fps_temp.argtypes = [fp, ci, ci, ci, ip, fp]
        L.s6d_oracle_fps_temp.restype = ci
        L.s6d_oracle_gather.argtypes = [fp, ip, ci, ci, ci, ci, fp]
        L.s6d_oracle_ball_query.argtypes = [fp, fp, ci, ci, ci, cf, ci, ip]
        L.s6d_oracle_group_points.argtypes = [fp, ip, ci, ci, ci, ci, ci, fp]
        L.s6d_oracle_opt_n_threads.argtypes = [ci]
        L.s6d_oracle_set_contraction.argtypes = [ci]
        L.s6d_oracle_set_contraction.restype = None
        for f in (L.s6d_oracle_fps, L.s6d_oracle_gather, L.s6d_oracle_ball_query,
                  L.s6d_oracle_group_points, L.s6d_oracle_opt_n_threads):
            f.restype = ci
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _np_f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), "must be a contiguous float tensor"
    return t.detach().cpu().numpy()


def _np_i32(t):
    assert t.dtype == torch.int32 and t.is_contiguous(), "must be a contiguous int tensor"
    return t.detach().cpu().numpy()


def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32; sampling.cpp:70-91."""
    p = _np_f32(points)
    B, N, _ = p.shape
    out = np.zeros((B, nsamples), dtype=np.int32)
    rc = lib().s6d_oracle_fps(_f(p), B, N, nsamples, _i(out))
    assert rc == 0, rc
    return torch.from_numpy(out)


def furthest_point_sampling_with_temp(points, nsamples):
    """-> (idx, temp): also the kernel's running minimum squared distances after the last selection."""
    p = _np_f32(points)
    B, N, _ = p.shape
    out = np.zeros((B, nsamples), dtype=np.int32)
    temp = np.zeros((B, N), dtype=np.float32)
    rc = lib().s6d_oracle_fps_temp(_f(p), B, N, nsamples, _i(out), _f(temp))
    assert rc == 0, rc
    return torch.from_numpy(out), torch.from_numpy(temp)


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M); sampling.cpp:18-43."""
    p, ix = _np_f32(points), _np_i32(idx)
    B, C, N = p.shape
    M = ix.shape[1]
    out = np.zeros((B, C, M), dtype=np.float32)
    rc = lib().s6d_oracle_gather(_f(p), _i(ix), B, C, N, M, _f(out))
    assert rc == 0, rc
    return torch.from_numpy(out)


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,M,3), (B,N,3) -> (B,M,nsample) i32; ball_query.cpp:13-37."""
    q, p = _np_f32(new_xyz), _np_f32(xyz)
    B, M, _ = q.shape
    N = p.shape[1]
    out = np.zeros((B, M, nsample), dtype=np.int32)
    rc = lib().s6d_oracle_ball_query(_f(q), _f(p), B, N, M, float(radius), int(nsample), _i(out))
    assert rc == 0, rc
    return torch.from_numpy(out)


def group_points(points, idx):
    """(B,C,N) f32, (B,M,S) i32 -> (B,C,M,S); group_points.cpp:14-38."""
    p, ix = _np_f32(points), _np_i32(idx)
    B, C, N = p.shape
    _, M, S = ix.shape
    out = np.zeros((B, C, M, S), dtype=np.float32)
    rc = lib().s6d_oracle_group_points(_f(p), _i(ix), B, C, N, M, S, _f(out))
    assert rc == 0, rc
    return torch.from_numpy(out)


def _unused(*a, **k):  # names that must exist on the pybind surface; training only
    raise NotImplementedError("training-only op; not on the inference hot path")


gather_points_grad = group_points_grad = three_nn = three_interpolate = three_interpolate_grad = _unused


class contraction:
    """``with contraction(mode):`` -- the spelling of a*a + b*b + c*c inside the distance (pn2_oracle.c header):
    0 nvcc default (the product's), 1 LLVM's, 2 none."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = lib().s6d_oracle_get_contraction()
        lib().s6d_oracle_set_contraction(self.mode)

    def __exit__(self, *a):
        lib().s6d_oracle_set_contraction(self.old)
