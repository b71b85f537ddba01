"""Boundary b2: the reference's OWN wrapper module pointnet2_utils.py, imported unmodified from
/root/reference, bound to ``sam6d_amd.pointnet2._ext`` in place of its CUDA pybind extension -- what a maintainer gets after the
one-line change of INTEGRATION.md.  furthest_point_sample / gather_operation / QueryAndGroup (pointnet2_utils.py:51-117,
:260-376) then run the gfx950 kernel sources (through the emulator here: no GPU in the build container) and are compared with
the C oracle.  Needs the reference tree: skipped on the GPU box."""
import importlib.util
import os
import sys
import types

import pytest
import torch

from oracle import pn2 as opn2

pytestmark = pytest.mark.ref
REF = os.environ.get("S6D_REFERENCE_ROOT", "/root/reference")
PN2_DIR = os.path.join(REF, "SAM-6D", "Pose_Estimation_Model", "model", "pointnet2")


def _reference_wrappers_on_the_shim():
    from sam6d_amd.pointnet2 import _ext as shim
    saved = {k: sys.modules.get(k) for k in ("pointnet2", "pointnet2._ext", "pytorch_utils")}
    pkg = types.ModuleType("pointnet2")
    pkg.__path__ = []
    pkg._ext = shim
    sys.modules["pointnet2"], sys.modules["pointnet2._ext"] = pkg, shim
    sys.path.insert(0, PN2_DIR)                                       # pointnet2_utils imports its sibling pytorch_utils
    try:
        spec = importlib.util.spec_from_file_location("ref_pointnet2_utils_on_shim", os.path.join(PN2_DIR, "pointnet2_utils.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(PN2_DIR)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert mod._ext is shim
    return mod


def test_reference_pointnet2_utils_runs_on_the_shim(emu):
    pu = _reference_wrappers_on_the_shim()
    g = torch.Generator().manual_seed(3)
    B, N, M, C = 2, 300, 40, 5
    xyz = torch.rand(B, N, 3, generator=g)
    feats = torch.randn(B, C, N, generator=g)
    # furthest_point_sample -> FurthestPointSampling.apply -> _ext.furthest_point_sampling (pointnet2_utils.py:51-78)
    idx = pu.furthest_point_sample(xyz, M)
    assert idx.dtype == torch.int32 and torch.equal(idx, opn2.furthest_point_sampling(xyz, M))
    # gather_operation -> _ext.gather_points (:81-117)
    got = pu.gather_operation(feats.contiguous(), idx)
    assert torch.equal(got, opn2.gather_points(feats.contiguous(), idx))
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    # QueryAndGroup -> ball_query + grouping_operation -> _ext.ball_query / _ext.group_points (:260-376)
    qg = pu.QueryAndGroup(0.25, 16, use_xyz=True)
    out = qg(xyz, new_xyz, feats.contiguous())
    bq = opn2.ball_query(new_xyz, xyz, 0.25, 16)
    gx = opn2.group_points(xyz.transpose(1, 2).contiguous(), bq) - new_xyz.transpose(1, 2).unsqueeze(-1)
    gf = opn2.group_points(feats.contiguous(), bq)
    assert out.shape == (B, 3 + C, M, 16) and torch.equal(out, torch.cat([gx, gf], dim=1))
    # the training-only names exist (the import surface of bindings.cpp:11-24) and say what they are when called
    with pytest.raises(NotImplementedError):
        pu.three_nn(new_xyz, xyz)
