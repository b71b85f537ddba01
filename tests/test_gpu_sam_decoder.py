"""GPU parity of the drop-in SAM prompt encoder + mask decoder vs the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import sam_decoder as osd
from sam6d_amd.utils import seeded
from tests import util
from tests.test_host_sam_decoder import build, case, run

pytestmark = pytest.mark.gpu


def _cuda(inp):
    return {k: v.cuda() for k, v in inp.items()}


@pytest.mark.parametrize("force_lib", [False, True])
def test_mini_fp32_vs_reference_golden(monkeypatch, force_lib):
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "fp32")
    g, c, cfg, inp = case("mini")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    with torch.no_grad():
        for tag, kw in (("", dict(points=(inp["points"], inp["labels"]))),
                        ("2", dict(points=(inp["points2"], inp["labels2"]), multi=False)),
                        ("_box", dict(boxes=inp["boxes"]))):
            s, mk, iou = run(m, inp["emb"], force_lib=force_lib, **kw)
            np.testing.assert_allclose(s.cpu().numpy(), g["mini_sparse" + tag], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(mk.cpu().numpy(), g["mini_masks" + tag], rtol=1e-3, atol=1e-4)
            np.testing.assert_allclose(iou.cpu().numpy(), g["mini_iou" + tag], rtol=1e-3, atol=1e-4)


def test_released_config_fp32_and_bf16_vs_reference_golden(monkeypatch):
    g, c, cfg, inp = case("sam")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "fp32")
    with torch.no_grad():
        _, mk, iou = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
    np.testing.assert_allclose(iou.cpu().numpy(), g["sam_iou"], rtol=1e-3, atol=1e-4)
    util.assert_digest_close(mk, g["sam_masks_sum"], g["sam_masks_smp"], 211, 1e-3, 1e-4, "low-res logits fp32")
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "bf16")
    with torch.no_grad():
        _, mk16, iou16 = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
    smp = mk16.float().cpu().reshape(-1)[::211].numpy()
    assert np.corrcoef(smp, g["sam_masks_smp"])[0, 1] > 0.999, np.corrcoef(smp, g["sam_masks_smp"])[0, 1]
    assert np.abs(smp - g["sam_masks_smp"]).mean() < 0.02 * np.abs(g["sam_masks_smp"]).mean() + 1e-3
    assert np.abs(iou16.cpu().numpy() - g["sam_iou"]).max() < 2e-2


def test_postprocess_matches_oracle():
    g, c, cfg, _ = case("mini")
    x = torch.from_numpy(g["mini_masks"][:3]).cuda()
    y = osd.postprocess_masks(x, cfg["img"], c["mini_input_size"], c["mini_orig"])      # torch ops on the device
    np.testing.assert_allclose(y.cpu().numpy(), g["mini_post"], rtol=1e-5, atol=1e-6)
