"""ctypes front-end of oracle/_ref/libpn2_ref_{off,fast}.so -- the REFERENCE's own PointNet++ CUDA kernels compiled for the host
by oracle/build_ref.py (TEST INFRASTRUCTURE; the libraries exist in the build container and travel to the GPU box as built
files, the sources they come from never enter the repo).  Same call shapes as oracle/pn2.py."""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def available(variant="off"):
    return os.path.exists(os.path.join(_HERE, "_ref", f"libpn2_ref_{variant}.so"))


def lib(variant="off"):
    if variant not in _LIBS:
        path = os.path.join(_HERE, "_ref", f"libpn2_ref_{variant}.so")
        if not os.path.exists(path):
            from . import build_ref
            build_ref.build()
        L = ctypes.CDLL(path)
        fp, ip, ci, cf = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_float
        L.pn2ref_fps.argtypes = [ci, ci, ci, fp, fp, ip]
        L.pn2ref_gather.argtypes = [ci, ci, ci, ci, fp, ip, fp]
        L.pn2ref_ball_query.argtypes = [ci, ci, ci, cf, ci, fp, fp, ip]
        L.pn2ref_group.argtypes = [ci, ci, ci, ci, ci, fp, ip, fp]
        for f in (L.pn2ref_fps, L.pn2ref_gather, L.pn2ref_ball_query, L.pn2ref_group):
            f.restype = None
        _LIBS[variant] = L
    return _LIBS[variant]


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def furthest_point_sampling(points, nsamples, variant="off", with_temp=False):
    """sampling.cpp:70-91: idxs = zeros, temp = full(1e10), then the kernel wrapper."""
    p = np.ascontiguousarray(points.numpy(), np.float32)
    B, N, _ = p.shape
    out = np.zeros((B, nsamples), np.int32)
    tmp = np.full((B, N), 1e10, np.float32)
    lib(variant).pn2ref_fps(B, N, nsamples, _f(p), _f(tmp), _i(out))
    return (torch.from_numpy(out), torch.from_numpy(tmp)) if with_temp else torch.from_numpy(out)


def gather_points(points, idx, variant="off"):
    p, ix = np.ascontiguousarray(points.numpy(), np.float32), np.ascontiguousarray(idx.numpy(), np.int32)
    B, C, N = p.shape
    out = np.zeros((B, C, ix.shape[1]), np.float32)
    lib(variant).pn2ref_gather(B, C, N, ix.shape[1], _f(p), _i(ix), _f(out))
    return torch.from_numpy(out)


def ball_query(new_xyz, xyz, radius, nsample, variant="off"):
    """ball_query.cpp:14-32: idx = zeros, then the kernel wrapper."""
    q, p = np.ascontiguousarray(new_xyz.numpy(), np.float32), np.ascontiguousarray(xyz.numpy(), np.float32)
    B, M, _ = q.shape
    out = np.zeros((B, M, nsample), np.int32)
    lib(variant).pn2ref_ball_query(B, p.shape[1], M, float(radius), nsample, _f(q), _f(p), _i(out))
    return torch.from_numpy(out)


def group_points(points, idx, variant="off"):
    p, ix = np.ascontiguousarray(points.numpy(), np.float32), np.ascontiguousarray(idx.numpy(), np.int32)
    B, C, N = p.shape
    _, npoints, ns = ix.shape
    out = np.zeros((B, C, npoints, ns), np.float32)
    lib(variant).pn2ref_group(B, C, N, npoints, ns, _f(p), _i(ix), _f(out))
    return torch.from_numpy(out)
