"""Host-logic test of the drop-in ISM scoring on CPU (library-op path) vs the reference golden."""
import ast

import numpy as np
import torch

from sam6d_amd.utils import synth
from tests import util


def test_frame_scoring_matches_reference_golden():
    from sam6d_amd.ism.scoring import FrameScorer, compute_iou
    g = util.golden("ism_scoring.npz")
    c = ast.literal_eval(str(g["case"]))
    inp = synth.ism_inputs(P=c["P"], O=c["O"], T=c["T"], seed=c["seed"])
    fs = FrameScorer(inp["ref_cls"], inp["ref_patch"], inp["poses"], inp["pointcloud"])
    np.testing.assert_allclose(fs.matching_config.metric(inp["qry_cls"], inp["ref_cls"]).numpy(), g["pairwise"],
                               atol=1e-6)
    out = fs.score(inp["qry_cls"], inp["qry_patch"], inp["masks"], inp["boxes"], inp["depth"], inp["K"])
    for k in ("sel", "pred_obj", "best_template", "image_uv"):
        assert np.array_equal(out[k].numpy(), g[k]), k
    for k in ("semantic", "appearance", "visible_ratio", "iou", "final"):
        np.testing.assert_allclose(torch.as_tensor(out[k]).numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    xyxy = torch.cat((out["image_uv"].min(1).values, out["image_uv"].max(1).values), -1).float()
    b = torch.from_numpy(g["boxes2"]).clone()
    b[0] += 10000
    assert compute_iou(xyxy, b) == 0.0  # quirk Q3
