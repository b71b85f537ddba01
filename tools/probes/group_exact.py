"""Is a FramePipeline group bit-identical with frame-by-frame calls, and if not, is it the PEM's INPUTS (pre-processing, template
rows) or the Net that differs?  Runs the mini pipeline under S6D_PEM_VIT_DTYPE = fp16 (the benched extractor: this library's own
GEMMs) and fp32 (library GEMMs), records the dict handed to the Net in both modes of calling and the outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_zz_pipeline import build_mini  # noqa: E402

dev = torch.device("cuda", 0)
for dt in ("fp16", "fp32"):
    os.environ["S6D_PEM_VIT_DTYPE"] = dt
    pipe, (img, depth, K, keys, rand_u) = build_mini(dev)
    img2, depth2 = img.flip(1).contiguous(), (depth + 0.02).contiguous()
    keys2, ru2 = keys.flip(0).contiguous(), rand_u.flip(0).contiguous()
    seen = []
    real = pipe._pem_forward

    def spy(ep):
        seen.append({k: v.clone() for k, v in ep.items()})
        return real(ep)
    pipe._pem_forward = spy
    single = [pipe(img, depth, K, keys, rand_u), pipe(img2, depth2, K, keys2, ru2)]
    group = pipe.run_group([(img, depth, K, keys, rand_u), (img2, depth2, K, keys2, ru2)])
    a, b, g = seen
    for k in g:
        cat = torch.cat([a[k], b[k]])
        print(f"{dt}: Net input {k:14s} group == cat(singles): {torch.equal(cat, g[k])}")
    for i, ((d1, p1), (d2, p2)) in enumerate(zip(single, group)):
        print(f"{dt}: frame {i}: scores equal {torch.equal(d1.scores, d2.scores)}",
              {k: f"{(p1[k].float() - p2[k].float()).abs().max().item():.3e}" for k in ("pred_R", "pred_t", "pred_pose_score")})
    # the Net alone on the group's dict, whole vs the two halves (eager)
    with torch.no_grad():
        n1 = a["pts"].shape[0]
        full = pipe.pem(dict(g))
        h1 = pipe.pem({k: v[:n1].contiguous() for k, v in g.items()})
        h2 = pipe.pem({k: v[n1:].contiguous() for k, v in g.items()})
    for k in ("init_R", "pred_R", "pred_t"):
        w = torch.cat([h1[k], h2[k]])
        print(f"{dt}: Net eager, {g['pts'].shape[0]} instances vs {n1} + {g['pts'].shape[0] - n1}: {k} equal {torch.equal(full[k], w)} max |d| {(full[k] - w).abs().max().item():.3e}")
    del pipe
    torch.cuda.empty_cache()
