"""``pointnet2._ext`` on MI355X: the nine names of the reference pybind module
(Pose_Estimation_Model/model/pointnet2/_ext_src/src/bindings.cpp:11-24).

The four forward ops used at inference run hand-written gfx950 kernels through the C ABI.
The five training / PointnetFPModule ops are not on the inference path:
their names exist so the import surface matches, and they raise if called.
"""
from ..ops import ball_query, furthest_point_sampling, gather_points, group_points  # noqa: F401


def _training_only(name):
    def f(*a, **k):
        raise NotImplementedError(f"pointnet2._ext.{name} is a training-only op; the MI355X build covers the "
                                  "per-frame inference hot path")
    f.__name__ = name
    return f


gather_points_grad = _training_only("gather_points_grad")
group_points_grad = _training_only("group_points_grad")
three_nn = _training_only("three_nn")
three_interpolate = _training_only("three_interpolate")
three_interpolate_grad = _training_only("three_interpolate_grad")
