"""Bodies of tests/test_gpu_zz_host_glue.py on the emulator: Detections NMS per object id through the (emulated) NMS kernel,
and the frame resize as integer tensor ops."""
from tests import test_gpu_zz_host_glue as T


def test_detections_nms_per_object_id_on_the_emulator(emu):
    T.test_detections_nms_per_object_id_on_the_device()


def test_frame_resize_body_on_the_emulator(emu, monkeypatch):
    import torch
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: self)
    T.test_frame_resize_on_the_device_is_pillow_exact()


def test_nonfinite_rows_kernel_on_the_emulator(emu):
    from tests import test_gpu_pem as TP
    TP.test_nonfinite_rows_kernel_flags_exactly_the_rows_with_inf_or_nan()
