"""Host-glue operations on the device (run last: they follow the kernel parity tests): Detections NMS per object id through
s6d_nms_f32, and the Pillow-exact frame resize as integer tensor ops on cuda:0."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_detections_nms_per_object_id_on_the_device():
    """Detections.apply_nms / apply_nms_per_object_id with the device NMS kernel (the default) reproduce the reference's
    Detections class on the golden frame (tests/golden/detections_ops.npz)."""
    import numpy as np

    from tests import util
    from tests.test_host_detections_ops import inputs
    g = util.golden("detections_ops.npz")

    def on_device():
        d = inputs(g)
        d.masks, d.boxes, d.scores, d.object_ids = d.masks.cuda(), d.boxes.cuda(), d.scores.cuda(), d.object_ids.cuda()
        return d
    d = on_device().apply_nms(0.5)
    np.testing.assert_array_equal(d.scores.cpu().numpy(), g["nms_scores"])
    d = on_device().apply_nms_per_object_id(0.25)
    np.testing.assert_array_equal(d.scores.cpu().numpy(), g["nms_obj_scores"])
    np.testing.assert_array_equal(d.object_ids.cpu().numpy(), g["nms_obj_ids"])
    np.testing.assert_array_equal(d.boxes.cpu().numpy(), g["nms_obj_boxes"])
    np.testing.assert_array_equal(d.masks.sum(dim=(1, 2)).cpu().numpy(), g["nms_obj_mask_sums"])


def test_frame_resize_on_the_device_is_pillow_exact():
    """sam/transforms.py integer resampler on cuda:0 against the reference-made golden (Pillow's pixels)."""
    import numpy as np

    from sam6d_amd.sam.transforms import ResizeLongestSide
    from tests import util
    g = util.golden("sam_transforms.npz")
    for tag in ("vga", "tless", "itodd", "tall"):
        out = ResizeLongestSide(int(g[tag + "_L"])).apply_image(torch.from_numpy(g[tag + "_img"]).cuda())
        assert out.is_cuda and out.dtype == torch.uint8
        np.testing.assert_array_equal(out.cpu().numpy(), g[tag + "_out"])
