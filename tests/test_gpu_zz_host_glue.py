"""Host-glue operations on the device (run last: they follow the kernel parity tests): Detections NMS per object id through
s6d_nms_f32, and the Pillow-exact frame resize as integer tensor ops on cuda:0."""
import os

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


def test_sequential_centroid_matches_numpy_row_order():
    """s6d_segment_seq_sum_f32 == numpy's add.reduce over axis 0, bit for bit, and with it the pre-processing agrees with the
    oracle at radii whose sphere cuts through dense points (where the float64-accumulated centroid does not)."""
    import numpy as np

    from oracle import pem_pre as opre
    from sam6d_amd import ops
    from sam6d_amd.pem import preprocess as pre
    from sam6d_amd.utils import synth
    g = torch.Generator().manual_seed(0)
    counts = torch.tensor([0, 1, 5, 511, 512, 513, 3000, 70001])
    x = (torch.randn(int(counts.sum()), 3, generator=g) * torch.tensor([0.3, 0.2, 0.1]) + torch.tensor([0.1, -0.2, 0.8]))
    start = torch.cumsum(counts, 0) - counts
    got = ops.segment_seq_sum(x.cuda(), start.cuda(), counts.cuda()).cpu().numpy()
    for i, (s0, c) in enumerate(zip(start.tolist(), counts.tolist())):
        want = np.add.reduce(x[s0:s0 + c].numpy(), axis=0) if c else np.zeros(3, np.float32)
        np.testing.assert_array_equal(got[i], want, err_msg=str(c))
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    radius = np.array([0.12, 0.03, 0.5, 0.12, 0.06, 0.2, 0.07, 0.01])
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), radius,
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]).cuda(), inp["depth"].cuda(), inp["K"], inp["masks"].cuda(),
                              torch.from_numpy(radius).cuda(), keys=inp["keys"].cuda(), **kw)
    assert out["kept"].cpu().tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].cpu().numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].cpu().numpy(), ref["rgb_choose"])


def test_sampler_kernel_equals_the_library_path_on_the_device(monkeypatch):
    """s6d_pem_sample_indices_f32 against the composite-key top-k it replaces, at frame sizes (480 x 640 keys per detection)."""
    from sam6d_amd import ops
    from sam6d_amd.pem import preprocess as pre
    g = torch.Generator().manual_seed(0)
    L, ns = 480 * 640, 2048
    n = torch.tensor([0, 1, 100, 2048, 2049, 3000, 10000, 100000, 307200, 5000])
    keys = torch.rand(len(n), L, generator=g)
    keys[5] = (keys[5] * 3000).floor() / 3000                       # ties: the position decides
    monkeypatch.setenv("S6D_PEM_SAMPLER", "library")
    want = pre._keyed_indices(n.cuda(), keys.cuda(), ns)
    idx, overflow = ops.pem_sample_indices(keys.cuda(), n.cuda(), ns)
    assert overflow.cpu().tolist() == [0] * len(n) and torch.equal(idx, want)


def test_kernel_path_of_the_preprocessing_on_the_device(monkeypatch):
    """The kernel path (the default on the device) on cuda:0 against the oracle loop (boundary-cutting radii included), the
    library-op path (S6D_PEM_PRE=library) against the same oracle, the chunked form of a many-detection frame, timing at 64
    detections."""
    import time

    import numpy as np

    from oracle import pem_pre as opre
    from sam6d_amd.pem import preprocess as pre
    from sam6d_amd.utils import synth
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(n_sample=2048, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    radius = np.array([0.12, 0.03, 0.5, 0.12, 0.06, 0.2, 0.07, 0.01])
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), radius,
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]).cuda(), inp["depth"].cuda(), inp["K"], inp["masks"].cuda(),
                              torch.from_numpy(radius).cuda(), keys=inp["keys"].cuda(), **kw)
    assert out["kept"].cpu().tolist() == ref["kept"].tolist()
    for k in ("bbox", "pts", "rgb_choose", "rgb"):
        np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k], err_msg=k)
    monkeypatch.setenv("S6D_PEM_PRE", "library")
    lib = pre.observed_inputs(torch.from_numpy(inp["image"]).cuda(), inp["depth"].cuda(), inp["K"], inp["masks"].cuda(),
                              torch.from_numpy(radius).cuda(), keys=inp["keys"].cuda(), **kw)
    monkeypatch.delenv("S6D_PEM_PRE")
    for k in ("kept", "bbox", "pts", "rgb_choose"):
        np.testing.assert_array_equal(lib[k].cpu().numpy(), ref[k], err_msg="library path: " + k)
    big = synth.pem_pre_inputs(P=64, seed=9)
    args = (torch.from_numpy(big["image"]).cuda(), big["depth"].cuda(), big["K"], big["masks"].cuda(), 0.15, big["keys"].cuda())
    whole = pre.observed_inputs(*args)
    monkeypatch.setattr(pre, "_SLOT_BYTES", 16 * 480 * 480 * 10)     # 10 detections per chunk
    parts = pre.observed_inputs(*args)
    monkeypatch.undo()
    for k in whole:
        assert torch.equal(whole[k], parts[k]), "chunked call: " + k
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        pre.observed_inputs(*args)
    torch.cuda.synchronize()
    print(f"PEM pre-processing, kernel path, 64 detections: {(time.time() - t0) / 3 * 1e3:.1f} ms")

