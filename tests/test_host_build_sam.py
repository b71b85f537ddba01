"""The Sam container / builders keep the reference's checkpoint surface: same state_dict keys, shapes and dtypes as
segment_anything/build_sam.py run unmodified (both built on the meta device: no memory, no arithmetic), for all three
encoder sizes; and the segmentor plugin keeps the reference's constructor."""
import inspect

import pytest
import torch

from oracle import refharness as rh
from sam6d_amd.ism import segmentor
from sam6d_amd.sam import build_sam

needs_ref = pytest.mark.skipif(not rh.available(), reason="reference tree not mounted")


@needs_ref
@pytest.mark.parametrize("name", ["vit_h", "vit_l", "vit_b", "default"])
def test_state_dict_surface_equals_the_reference(name):
    ref_mod = rh.sam_builder()
    with torch.device("meta"):
        ours = build_sam.sam_model_registry[name]()
        ref = ref_mod.sam_model_registry[name]()
    a = {k: (tuple(v.shape), v.dtype) for k, v in ours.state_dict().items()}
    b = {k: (tuple(v.shape), v.dtype) for k, v in ref.state_dict().items()}
    assert a == b and len(a) > 300
    assert list(ours.state_dict()) == list(ref.state_dict())                   # same order as well
    assert ours.image_encoder.img_size == ref.image_encoder.img_size == 1024
    assert ours.mask_threshold == ref.mask_threshold and ours.image_format == ref.image_format
    assert "pixel_mean" not in a and tuple(ours.pixel_mean.shape) == tuple(ref.pixel_mean.shape)   # non-persistent buffers
    assert not ours.training


@pytest.mark.parametrize("name", ["vit_h", "vit_l", "vit_b"])
def test_state_dict_surface_equals_the_committed_golden(name):
    """Same check against tests/golden/sam_state_dict.json (made by oracle/gen_golden.py sam_state_dict), for hosts
    without the reference tree."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "sam_state_dict.json")) as f:
        want = json.load(f)[name]
    with torch.device("meta"):
        ours = build_sam.sam_model_registry[name]()
    assert [[k, list(v.shape)] for k, v in ours.state_dict().items()] == want


def test_registry_and_segmentor_signature():
    assert set(build_sam.sam_model_registry) == {"default", "vit_h", "vit_l", "vit_b"}
    assert build_sam.build_sam is build_sam.build_sam_vit_h
    sig = inspect.signature(segmentor.CustomSamAutomaticMaskGenerator.__init__)
    assert list(sig.parameters)[1:] == ["sam", "min_mask_region_area", "points_per_batch", "stability_score_thresh",
                                        "box_nms_thresh", "crop_overlap_ratio", "segmentor_width_size", "pred_iou_thresh"]
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert d == dict(min_mask_region_area=0, points_per_batch=64, stability_score_thresh=0.85, box_nms_thresh=0.7,
                     crop_overlap_ratio=512 / 1500, segmentor_width_size=None, pred_iou_thresh=0.88)
    with pytest.raises(NotImplementedError):
        segmentor.CustomSamAutomaticMaskGenerator(None, min_mask_region_area=10)
    assert segmentor.pretrained_weight_dict["vit_h"] == "sam_vit_h_4b8939.pth"


def test_generate_masks_glue_with_stand_in_models(monkeypatch):
    """CustomSamAutomaticMaskGenerator.generate_masks on a host without a device: encoder / proposal generator replaced by
    recording stand-ins; the resize chain, the arguments handed down and the resize back are the real code."""
    import numpy as np

    from sam6d_amd.sam import amg
    from sam6d_amd.sam.transforms import ResizeLongestSide, pil_bilinear_resize_u8
    seen = {}

    class Enc:
        img_size = 128

        def __call__(self, x):
            seen["enc_in"] = x
            return torch.zeros(1, 4, 8, 8)

    class FakeSam:
        device = torch.device("cpu")
        mask_threshold = 0.0
        image_encoder, prompt_encoder, mask_decoder = Enc(), "PE", "MD"

        def preprocess(self, x):
            seen["pre_in"] = x.clone()
            return x

    def fake_proposals(pe, md, emb, size, img_size, **kw):
        seen["args"] = (pe, md, tuple(emb.shape), size, img_size, kw)
        m = torch.zeros(2, *size, dtype=torch.bool)
        m[0, 2:20, 3:30] = True
        m[1, 10:40, 40:60] = True
        return dict(masks=m, boxes=torch.tensor([[3, 2, 29, 19], [40, 10, 59, 39]]))
    monkeypatch.setattr(segmentor.amg, "generate_proposals", fake_proposals)
    img = np.random.default_rng(1).integers(0, 256, (54, 72, 3), dtype=np.uint8)
    # ---- with segmentor_width_size: frame shrunk to 48 x 64, proposals found there, results resized back ---------------
    gen = segmentor.CustomSamAutomaticMaskGenerator(FakeSam(), segmentor_width_size=64, stability_score_thresh=0.5)
    out = gen.generate_masks(img)
    small = pil_bilinear_resize_u8(torch.from_numpy(img), (48, 64))
    want = ResizeLongestSide(128).apply_image(small).permute(2, 0, 1)[None].float()
    assert torch.equal(seen["pre_in"], want) and want.shape == (1, 3, 96, 128)
    pe, md, es, size, isz, kw = seen["args"]
    assert (pe, md, es, size, isz) == ("PE", "MD", (1, 4, 8, 8), (48, 64), 128)
    assert kw == dict(points_per_side=32, points_per_batch=64, mask_threshold=0.0, pred_iou_thresh=0.88,
                      stability_score_thresh=0.5, stability_score_offset=1.0, box_nms_thresh=0.7)
    assert out["masks"].shape == (2, 54, 72) and out["masks"].dtype == torch.float32 and out["boxes"].dtype == torch.float32
    m_ref, b_ref = amg.postprocess_resize(fake_proposals(0, 0, torch.zeros(1), (48, 64), 0)["masks"],
                                          torch.tensor([[3, 2, 29, 19], [40, 10, 59, 39]]), (54, 72), 64)
    assert torch.equal(out["masks"], m_ref) and torch.equal(out["boxes"], b_ref)
    # ---- without: proposals on the frame itself, bool masks and integer boxes as found ----------------------------------
    out = segmentor.CustomSamAutomaticMaskGenerator(FakeSam()).generate_masks(torch.from_numpy(img))
    assert seen["args"][3] == (54, 72) and out["masks"].dtype == torch.bool and out["boxes"].dtype == torch.int64
    assert seen["pre_in"].shape == (1, 3, 96, 128)
    with pytest.raises(ValueError):
        gen.generate_masks(img.astype(np.float32))
