"""Where the group-batched descriptor path spends its time: CustomDINOv2.forward x 8 frames against forward_frames, and the pieces
of forward_frames (crops, concatenation, ViT batches, split)."""
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import frame_demo  # noqa: E402
from sam6d_amd.ism.dinov2 import plan_chunks  # noqa: E402
from sam6d_amd.utils import synth  # noqa: E402

dev = torch.device("cuda", 0)
pipe, args = frame_demo.build(dev, 1024, 10, sync_stages=False)
desc = pipe.desc
frame = synth.pem_pre_inputs(P=128, seed=3)
img = torch.from_numpy(frame["image"]).to(dev)
masks = frame["masks"].to(dev).float()
boxes = synth.dinov2_inputs(P=128, seed=3)["boxes"].to(dev)
pr = SimpleNamespace(masks=masks, boxes=boxes)


def tm(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / n


print(f"8 x forward (128 crops each): {tm(lambda: [desc(img, pr) for _ in range(8)]):.2f} ms")
print(f"forward_frames (8 frames): {tm(lambda: desc.forward_frames([img] * 8, [pr] * 8)):.2f} ms")
print(f"8 x crops: {tm(lambda: [desc._crops(img, pr.masks, pr.boxes, True, True) for _ in range(8)]):.2f} ms")
crops = [desc._crops(img, pr.masks, pr.boxes, True, True) for _ in range(8)]
print(f"cat: {tm(lambda: (torch.cat([c[0] for c in crops]), torch.cat([c[1] for c in crops]))):.2f} ms")
rgbs, mk = torch.cat([c[0] for c in crops]), torch.cat([c[1] for c in crops])
for c in sorted(set(plan_chunks(1024))):
    print(f"compute_cls_and_patch_features({c} crops): {tm(lambda: desc.compute_cls_and_patch_features(rgbs[:c], mk[:c])):.2f} ms")
print(f"compute_cls_and_patch_features(128 crops): {tm(lambda: desc.compute_cls_and_patch_features(rgbs[:128], mk[:128])):.2f} ms")
