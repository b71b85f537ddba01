"""The ISM's segmentor plugin end to end on one MI355X (run on the GPU box): sam6d_amd.ism.segmentor.
CustomSamAutomaticMaskGenerator.generate_masks on a synthetic 480x640 frame with a seeded ViT-H Sam -- frame upload, resize,
encoder, 1024 prompts, filters, NMS, resize back.  Seeded weights give meaningless masks, so the thresholds are lowered for
them (as in tools/frame_demo.py); the number printed is the wall time per frame, not a quality statement."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sam6d_amd.ism.segmentor import CustomSamAutomaticMaskGenerator  # noqa: E402
from sam6d_amd.sam.build_sam import build_sam_vit_h  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402

dev = torch.device("cuda", 0)
sam = build_sam_vit_h()
seeded.load_seeded(sam.image_encoder, 3)
seeded.load_seeded(sam.prompt_encoder, 2)
seeded.load_seeded(sam.mask_decoder, 2)
sam = sam.to(dev)
gen = CustomSamAutomaticMaskGenerator(sam, points_per_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 1024,
                                      segmentor_width_size=640, stability_score_thresh=0.3, pred_iou_thresh=0.09)
gen.stability_score_offset = 0.02
img = synth.pem_pre_inputs(P=4, seed=3)["image"]                     # (480,640,3) uint8 numpy, as the reference passes it
for it in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = gen.generate_masks(img)
    torch.cuda.synchronize()
    print(f"frame {it}: {(time.perf_counter() - t) * 1e3:.1f} ms, {out['masks'].shape[0]} masks, masks {tuple(out['masks'].shape)} "
          f"{out['masks'].dtype}, boxes {out['boxes'].dtype}", flush=True)
