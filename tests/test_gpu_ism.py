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


def test_ism_kernels_individually_vs_oracle():
    """Each fused ISM kernel against the CPU oracle on odd, non-tile-aligned sizes."""
    from oracle import ism as oism
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(3)
    # cosine: P, R not multiples of 16
    q, r = torch.randn(37, 1024, generator=g), torch.randn(3, 45, 1024, generator=g)
    ref = oism.pairwise_similarity(q, r)
    out = ops.pairwise_cosine(q.cuda(), r.reshape(-1, 1024).cuda()).cpu().view(37, 3, 45)
    assert (out - ref).abs().max() < 2e-6
    # semantic select vs torch (avg_5 / max / mean)
    sc = torch.rand(37, 3, 45, generator=g)
    for topk in (5, 1, 45):
        bs, bo, bt = (t.cpu() for t in ops.semantic_select(sc.cuda(), topk))
        per = torch.topk(sc, k=topk, dim=-1)[0].mean(-1)
        es, eo = per.max(-1)
        assert torch.equal(bo.long(), eo) and (bs - es).abs().max() < 1e-6
        assert torch.equal(bt.long(), torch.gather(sc.argmax(-1), 1, eo[:, None])[:, 0])
    # patch scores: N1 = 200 rows (not a multiple of 64), N2 = 250 cols, zeroed rows/cols
    S, N1, N2, C = 5, 200, 250, 256
    store = torch.nn.functional.normalize(torch.randn(2, 3, N2, C, generator=g), dim=-1)
    store = store * (torch.rand(2, 3, N2, 1, generator=g) > 0.3)
    obj = torch.tensor([0, 1, 1, 0, 1])
    tm = torch.tensor([2, 0, 1, 1, 2])
    qp = torch.nn.functional.normalize(0.6 * store[obj, tm][:, :N1] + 0.4 * torch.randn(S, N1, C, generator=g), dim=-1)
    qp = qp * (torch.rand(S, N1, 1, generator=g) > 0.3)
    ea, eref = oism.appearance_score(qp, store, obj, tm)
    er = oism.visible_ratio(qp, eref, 0.5)
    a, rr = ops.patch_scores(qp.cuda(), store.cuda(), obj.int().cuda(), tm.int().cuda(), 0.5)
    assert (a.cpu() - ea).abs().max() < 2e-6 and (rr.cpu() - er).abs().max() < 2e-6
