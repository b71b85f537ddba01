"""ADD / ADD-S pose-error metrics.

The reference computes no accuracy metric in-repo (outputs go to the external bop_toolkit);
the parity contract (BASELINE.json: identical ADD(-S) recall on identical inputs) therefore
needs an implementation applied identically to both sides.  Standard definitions:
  ADD   = mean_x || (R x + t) - (R* x + t*) ||
  ADD-S = mean_x min_y || (R x + t) - (R* y + t*) ||      (symmetric objects,
          cf. Pose_Estimation_Model/utils/bop_object_utils.py:81-86 symmetry_flag)
A pose is correct when the error is below ``frac`` x object diameter (default 10 %).
"""
import torch


def _transform(R, t, pts):
    return pts @ R.transpose(-1, -2) + t.unsqueeze(-2)


def add_error(R, t, R_gt, t_gt, model_pts):
    return (_transform(R, t, model_pts) - _transform(R_gt, t_gt, model_pts)).norm(dim=-1).mean(dim=-1)


def adds_error(R, t, R_gt, t_gt, model_pts):
    a, b = _transform(R, t, model_pts), _transform(R_gt, t_gt, model_pts)
    return torch.cdist(a, b).min(dim=-1)[0].mean(dim=-1)


def diameter(model_pts):
    return torch.cdist(model_pts, model_pts).flatten(-2).max(dim=-1)[0]


def add_recall(R, t, R_gt, t_gt, model_pts, diam=None, frac=0.1, symmetric=False):
    """Returns (recall in [0,1], per-instance boolean mask)."""
    diam = diameter(model_pts) if diam is None else torch.as_tensor(diam, dtype=model_pts.dtype)
    err = (adds_error if symmetric else add_error)(R, t, R_gt, t_gt, model_pts)
    ok = err < frac * diam
    return ok.float().mean().item(), ok
