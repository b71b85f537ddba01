"""GPU parity of the ISM scoring chain vs the reference golden."""
import ast

import numpy as np
import pytest
import torch

from sam6d_amd.utils import synth
from tests import util

pytestmark = pytest.mark.gpu


def test_frame_scoring_vs_reference_golden():
    from sam6d_amd.ism.scoring import FrameScorer
    g = util.golden("ism_scoring.npz")
    c = ast.literal_eval(str(g["case"]))
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in
           synth.ism_inputs(P=c["P"], O=c["O"], T=c["T"], seed=c["seed"]).items()}
    fs = FrameScorer(inp["ref_cls"], inp["ref_patch"], inp["poses"], inp["pointcloud"])
    np.testing.assert_allclose(fs.matching_config.metric(inp["qry_cls"], inp["ref_cls"]).cpu().numpy(),
                               g["pairwise"], atol=2e-6)
    out = fs.score(inp["qry_cls"], inp["qry_patch"], inp["masks"], inp["boxes"], inp["depth"], inp["K"])
    for k in ("sel", "pred_obj", "best_template"):
        assert np.array_equal(out[k].cpu().numpy(), g[k]), k
    d = np.abs(out["image_uv"].cpu().numpy() - g["image_uv"])
    assert d.max() <= 1 and (d > 0).mean() < 1e-3          # float->int truncation at pixel borders
    for k, tol in (("semantic", 1e-5), ("appearance", 1e-5), ("visible_ratio", 1e-5), ("iou", 2e-2), ("final", 1e-2)):
        np.testing.assert_allclose(torch.as_tensor(out[k]).cpu().numpy(), g[k], rtol=0, atol=tol, err_msg=k)
