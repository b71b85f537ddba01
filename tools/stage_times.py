"""Per-stage wall times of the MI355X PEM path (diagnostic; run on the GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sam6d_amd.pem import pose_estimation_model as pm  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402


def timed(name, fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.time() - t0) / n * 1e3:9.2f} ms", flush=True)
    return out


def main(B=32):
    net = pm.Net(pm.default_cfg()).eval()
    seeded.load_seeded(net, 1)
    net = net.cuda()
    inp = {k: v.cuda() for k, v in synth.pem_inputs(B, seed=1).items()}
    ru = synth.coarse_uniforms(B, 2).cuda()
    with torch.no_grad():
        fe = net.feature_extraction
        timed("vit tokens_up", lambda: fe.rgb_net.tokens_up(inp["rgb"]))
        dpm, dfm, dpo, dfo, rad = timed("feature_extraction", lambda: fe(inp))
        spm, sfm, im = timed("fps+gather", lambda: pm.sample_pts_feats(dpm, dfm, 196))
        spo, sfo, io = pm.sample_pts_feats(dpo, dfo, 196)
        bg = torch.full((B, 1, 3), 100.0, device="cuda")
        gm = timed("geo_embedding", lambda: net.geo_embedding(torch.cat([bg, spm], 1)))
        go = net.geo_embedding(torch.cat([bg, spo], 1))
        ep = dict(model=inp["model"], coarse_rand_u=ru)
        ep = timed("coarse matching", lambda: net.coarse_point_matching(spm, sfm, gm, spo, sfo, go, rad, dict(ep)))
        timed("fine matching", lambda: net.fine_point_matching(dpm, dfm, gm, im, dpo, dfo, go, io, rad, dict(ep)))
        timed("PE", lambda: net.fine_point_matching.PE(dpo))
        timed("Net.forward", lambda: net(dict(inp, coarse_rand_u=ru)))
    print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
