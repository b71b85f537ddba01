"""Which module of the PEM Net gives an instance different bits when the batch around it changes?  Forward hooks on every
submodule; the Net runs on 6 instances at once and on instances [0,2) + [2,6); the first modules (in call order) whose outputs differ
row-wise are printed.  Usage: python tools/probes/batch_invariance.py [fp32|fp16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    os.environ["S6D_PEM_VIT_DTYPE"] = sys.argv[1]
from sam6d_amd.pem import pose_estimation_model as pm  # noqa: E402
from sam6d_amd.utils import seeded, synth  # noqa: E402

dev = torch.device("cuda", 0)
net = seeded.load_seeded(pm.Net(pm.default_cfg()).eval(), 1).to(dev)
pin = synth.pem_inputs(6, seed=3)
keys_b = ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo", "coarse_rand_u")
ep = {k: v.to(dev) for k, v in pin.items() if torch.is_tensor(v)}
ep["coarse_rand_u"] = synth.coarse_uniforms(6, 5).to(dev)
log = []


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in flat(x)]
    if isinstance(o, dict):
        return [t for x in o.values() for t in flat(x)]
    return []


def hook(name):
    def f(mod, inp, out):
        log.append((name, [t.detach().clone() for t in flat(out) if t.is_floating_point()]))
    return f


for name, m in net.named_modules():
    if name:
        m.register_forward_hook(hook(name))


def run(sl):
    log.clear()
    with torch.no_grad():
        out = net({k: ep[k][sl].contiguous() for k in keys_b})
    return list(log), out


full, of = run(slice(0, 6))
a, oa = run(slice(0, 2))
b, ob = run(slice(2, 6))
assert [n for n, _ in full] == [n for n, _ in a] == [n for n, _ in b]
shown = 0
for (name, tf), (_, ta), (_, tb) in zip(full, a, b):
    for i, (x, y, z) in enumerate(zip(tf, ta, tb)):
        if x.shape[:1] != (6,) or y.shape[:1] != (2,):
            continue
        w = torch.cat([y, z])
        if not torch.equal(x, w):
            print(f"DIFFERS  {name}[{i}] {tuple(x.shape)} max |d| {(x - w).abs().max().item():.3e}")
            shown += 1
    if shown >= 12:
        break
for k in ("init_R", "init_t", "pred_R", "pred_t", "pred_pose_score"):
    w = torch.cat([oa[k], ob[k]])
    print(f"output {k}: equal {torch.equal(of[k], w)} max |d| {(of[k] - w).abs().max().item():.3e}")
