"""PEM per-detection pre-processing on the device against the oracle's per-detection loop."""
import time

import numpy as np
import pytest
import torch

from oracle import pem_pre as opre
from sam6d_amd.pem import preprocess as pre
from sam6d_amd.utils import synth

pytestmark = pytest.mark.gpu


def test_batched_preprocessing_on_device_matches_oracle_loop():
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(radius=0.12, n_sample=2048, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(),
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]).cuda(), inp["depth"].cuda(), inp["K"], inp["masks"].cuda(),
                              keys=inp["keys"].cuda(), **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["bbox"].cpu().numpy(), ref["bbox"])
    np.testing.assert_array_equal(out["rgb_choose"].cpu().numpy(), ref["rgb_choose"])
    np.testing.assert_array_equal(out["pts"].cpu().numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb"].cpu().numpy(), ref["rgb"])


def test_frame_of_many_detections_runs_in_milliseconds():
    inp = synth.pem_pre_inputs(P=64, seed=9)
    args = (torch.from_numpy(inp["image"]).cuda(), inp["depth"].cuda(), inp["K"], inp["masks"].cuda(), 0.15,
            inp["keys"].cuda())
    out = pre.observed_inputs(*args)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = pre.observed_inputs(*args)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 3 * 1e3
    M = out["pts"].shape[0]
    assert M >= 48 and out["pts"].shape == (M, 2048, 3) and out["rgb"].shape == (M, 3, 224, 224)
    assert torch.isfinite(out["pts"]).all() and (out["rgb_choose"] >= 0).all() and (out["rgb_choose"] < 224 * 224).all()
    print(f"PEM pre-processing, 64 detections of one 480x640 frame: {ms:.1f} ms")
    assert ms < 200
