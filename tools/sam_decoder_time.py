"""Stage timing of the SAM prompt encoder + mask decoder (section 8f-2) at the automatic-mask-generator shape:
1024 point prompts per frame against one (1,256,64,64) image embedding (run on the GPU box).
usage: sam_decoder_time.py [n_prompts] [chunk]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from sam6d_amd.utils import seeded, synth  # noqa: E402
from sam6d_amd.sam.mask_decoder import build_sam_decoder  # noqa: E402


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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = dict(dim=256, emb=64, img=1024)
    m = seeded.load_seeded(build_sam_decoder(), 1).cuda()
    inp = {k: v.cuda() for k, v in synth.sam_decoder_inputs(cfg, n, 3).items()}

    def frame(lib):
        outs = []
        with torch.no_grad():
            for a in range(0, n, chunk):
                s, d = m.prompt_encoder(points=(inp["points"][a:a + chunk], inp["labels"][a:a + chunk]), boxes=None, masks=None)
                if lib:
                    d = d.contiguous()
                outs.append(m.mask_decoder(image_embeddings=inp["emb"], image_pe=m.prompt_encoder.get_dense_pe(),
                                           sparse_prompt_embeddings=s, dense_prompt_embeddings=d, multimask_output=True))
        return outs
    for dt in ("bf16", "fp32"):
        os.environ["S6D_SAM_DECODER_DTYPE"] = dt; __import__("sam6d_amd.policy").policy.reload()
        for lib in (False, True):
            if lib and chunk > 64:
                continue                          # the as-written path materialises (chunk,4096,256) repeats: keep it small
            ms = ev(lambda: frame(lib), 3)
            print(f"{dt} {'reference op sequence' if lib else 'restructured'}: {n} prompts, chunk {chunk}: {ms:.2f} ms/frame "
                  f"-> {3.6e9 * n / ms / 1e9:.0f} TFLOP/s as-written-equivalent", flush=True)
    # mask post-processing of one batch of 256 prompts x 3 masks -> 480x640 frame
    from sam6d_amd import ops  # noqa: E402
    low = synth.sam_lowres_logits(4, 3, 256, 1).cuda().repeat(64, 1, 1, 1)           # (256,3,256,256)
    ms = ev(lambda: ops.sam_mask_post(low, 1024, (768, 1024), (480, 640), 0.0, 1.0), 5)
    print(f"mask post-processing (fused), 768 masks -> 480x640: {ms:.3f} ms ({768 * 480 * 640 / ms / 1e6:.1f} Gpixel/s); "
          f"x4 batches per 1024-prompt frame = {4 * ms:.2f} ms", flush=True)
    # the whole embedding -> proposals stage (grid, 1024 prompts, post-processing, filters, NMS); thresholds chosen so that
    # about half of the 3072 masks of these seeded weights survive the filters (worst-ish case for the NMS)
    from sam6d_amd.sam import amg  # noqa: E402
    os.environ["S6D_SAM_DECODER_DTYPE"] = "bf16"; __import__("sam6d_amd.policy").policy.reload()
    kw = dict(pred_iou_thresh=0.05, stability_score_thresh=0.3, stability_score_offset=0.02, points_per_batch=chunk)
    out = amg.generate_proposals(m.prompt_encoder, m.mask_decoder, inp["emb"], (480, 640), **kw)
    ms = ev(lambda: amg.generate_proposals(m.prompt_encoder, m.mask_decoder, inp["emb"], (480, 640), **kw), 3)
    print(f"generate_proposals: 1024 prompts -> {out['masks'].shape[0]} proposals after filters + NMS: {ms:.2f} ms/frame")
