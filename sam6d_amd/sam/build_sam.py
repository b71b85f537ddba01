"""The ``Sam`` container and its builders (segment_anything/build_sam.py:14-107, modeling/sam.py:18-52): image encoder +
prompt encoder + mask decoder under the attribute names -- hence state_dict keys -- of the published checkpoints
(``sam_vit_h_4b8939.pth`` loads with ``strict=True``; pixel_mean / pixel_std are non-persistent buffers as in the
reference).  ``Sam.forward`` (the batched end-to-end call, modeling/sam.py:54-131) is not on the SAM-6D path -- the ISM
drives the three sub-modules through its mask generator (sam6d_amd/ism/segmentor.py) -- and is not provided.
"""
from functools import partial

import torch
from torch import nn

from .image_encoder import PIXEL_MEAN, PIXEL_STD, ImageEncoderViT
from .image_encoder import preprocess as _preprocess
from .mask_decoder import MaskDecoder, PromptEncoder, TwoWayTransformer

# encoder width, depth, heads, global-attention block indexes (build_sam.py:14-44)
ENCODERS = {
    "vit_h": (1280, 32, 16, (7, 15, 23, 31)),
    "vit_l": (1024, 24, 16, (5, 11, 17, 23)),
    "vit_b": (768, 12, 12, (2, 5, 8, 11)),
}
PROMPT_DIM, IMAGE_SIZE, PATCH = 256, 1024, 16


class Sam(nn.Module):
    mask_threshold = 0.0
    image_format = "RGB"

    def __init__(self, image_encoder, prompt_encoder, mask_decoder, pixel_mean=PIXEL_MEAN, pixel_std=PIXEL_STD):
        super().__init__()
        self.image_encoder, self.prompt_encoder, self.mask_decoder = image_encoder, prompt_encoder, mask_decoder
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self._mean_std = (tuple(float(v) for v in pixel_mean), tuple(float(v) for v in pixel_std))   # host copies: no sync per frame

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess(self, x):
        """Normalise colours and zero-pad to the square input (modeling/sam.py:164-174), one fused kernel."""
        return _preprocess(x, self.image_encoder.img_size, *self._mean_std)

    def forward(self, *a, **k):
        raise NotImplementedError("Sam.forward (batched prompts end to end) is not on the SAM-6D path; use "
                                  "sam6d_amd.ism.segmentor.CustomSamAutomaticMaskGenerator or the sub-modules")


def _build_sam(model_type, checkpoint=None):
    width, depth, heads, global_idx = ENCODERS[model_type]
    grid = IMAGE_SIZE // PATCH
    sam = Sam(
        ImageEncoderViT(depth=depth, embed_dim=width, img_size=IMAGE_SIZE, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                        num_heads=heads, patch_size=PATCH, qkv_bias=True, use_rel_pos=True, global_attn_indexes=global_idx,
                        window_size=14, out_chans=PROMPT_DIM),
        PromptEncoder(embed_dim=PROMPT_DIM, image_embedding_size=(grid, grid), input_image_size=(IMAGE_SIZE, IMAGE_SIZE),
                      mask_in_chans=16),
        MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=PROMPT_DIM, mlp_dim=2048, num_heads=8),
                    transformer_dim=PROMPT_DIM, iou_head_depth=3, iou_head_hidden_dim=256)).eval()
    if checkpoint is not None:
        with open(checkpoint, "rb") as f:
            sam.load_state_dict(torch.load(f, map_location="cpu"))
    return sam


def build_sam_vit_h(checkpoint=None):
    return _build_sam("vit_h", checkpoint)


def build_sam_vit_l(checkpoint=None):
    return _build_sam("vit_l", checkpoint)


def build_sam_vit_b(checkpoint=None):
    return _build_sam("vit_b", checkpoint)


build_sam = build_sam_vit_h
sam_model_registry = {"default": build_sam_vit_h, "vit_h": build_sam_vit_h, "vit_l": build_sam_vit_l, "vit_b": build_sam_vit_b}
