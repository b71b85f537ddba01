"""Time the fine-matching head at the benched batch (B = 32, 2049 x 2049, C = 256): the fused similarity + assignment kernels
(s6d_fine_match_f32) against round 1's path (library bmm writing the (B,2049,2049) matrix + s6d_fine_assign_f32 streaming it
three times), and check the two against each other.   python tools/fine_time.py [B]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sam6d_amd import ops  # noqa: E402


def event_ms(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    g = torch.Generator(device="cuda").manual_seed(0)
    M = 2049
    f1 = torch.randn(B, M, 256, generator=g, device="cuda")
    perm = torch.randperm(M, generator=g, device="cuda")
    f2 = f1[:, perm] + 0.4 * torch.randn(B, M, 256, generator=g, device="cuda")
    pts2 = torch.randn(B, M - 1, 3, generator=g, device="cuda")
    F = torch.nn.functional

    def old():
        atten = F.normalize(f1, dim=2) @ F.normalize(f2, dim=2).transpose(1, 2) / 0.1
        return ops.fine_assign(atten, pts2)

    def new():
        return ops.fine_match(f1, f2, pts2, 0.1)
    po, wo, lo = old()
    pn, wn, ln = new()
    res = dict(B=B, label_mismatches=int((lo != ln).sum().item()), pred_max_diff=float((po - pn).abs().max().item()),
               wsum_max_rel=float(((wo - wn).abs() / wo.abs().clamp(min=1e-6)).max().item()),
               old_ms=round(event_ms(old), 4), new_ms=round(event_ms(new), 4))
    atten = F.normalize(f1, dim=2) @ F.normalize(f2, dim=2).transpose(1, 2) / 0.1
    res["old_assign_only_ms"] = round(event_ms(lambda: ops.fine_assign(atten, pts2)), 4)
    # algorithmic bytes of the fused path: the two feature sets read once + outputs
    alg = B * (2 * M * 256 * 4 + (M - 1) * 5 * 4)
    res["algorithmic_MB"] = round(alg / 1e6, 2)
    res["matrix_MB_avoided"] = round(B * M * M * 4 / 1e6, 1)
    res["mfma_tflops_equiv"] = round(3 * 3 * 2.0 * B * 2080 * 2048 * 256 / (res["new_ms"] * 1e-3) / 1e12, 1)
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fine_time.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
