"""Stage timing of the DINOv2 descriptor path (section 8f-1) at the frame shape: P proposals on a 480x640 frame ->
fused crops -> ViT-L/14 descriptors (run on the GPU box).  usage: dinov2_time.py [P] [chunk]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sam6d_amd.ism import dinov2 as pd  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402


def ev(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    inp = synth.dinov2_inputs(P=P, seed=2)
    m = seeded.load_seeded(pd._make_dinov2_model(arch_name="vit_large").eval(), 1).cuda()
    o = pd.CustomDINOv2.__new__(pd.CustomDINOv2)
    torch.nn.Module.__init__(o)
    o.model, o.patch_size, o.validpatch_thresh, o.chunk_size, o.proposal_size = m, 14, 0.5, chunk, 224
    masks, boxes = inp["masks"].cuda(), inp["boxes"].cuda()
    import types
    props = types.SimpleNamespace(masks=masks, boxes=boxes)
    t0 = time.time()
    o.forward(inp["image"], props)
    torch.cuda.synchronize()
    print(f"first call {time.time() - t0:.2f} s", flush=True)
    crop_ms = ev(lambda: o._crops(inp["image"], masks, boxes, True, True), 10)
    rgbs, pm = o._crops(inp["image"], masks, boxes, True, True)
    vit_ms = ev(lambda: o.compute_cls_and_patch_features(rgbs, pm) if P <= chunk else
                [o.compute_cls_and_patch_features(rgbs[a:b], pm[a:b]) for a, b in o._chunks(P)], 5)
    all_ms = ev(lambda: o.forward(inp["image"], props), 5)
    flop = 162.0e9 * P          # 162.0 GFLOP per 224^2 crop
    crop_bytes = P * 4 * 224 * 224 * 4.0
    print(f"P={P} chunk={chunk}: crops {crop_ms:.3f} ms ({crop_bytes / crop_ms / 1e6:.0f} GB/s written), "
          f"ViT-L/14 {vit_ms:.2f} ms ({flop / vit_ms / 1e9:.0f} TFLOP/s), forward() {all_ms:.2f} ms "
          f"-> {1e3 / all_ms:.1f} frames/s of descriptor extraction")
