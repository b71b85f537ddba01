"""Deterministic, checkpoint-free weights for parity tests and benchmarks.

No SAM / DINOv2 / PEM checkpoint is reachable offline, so every parity test and
bench run uses weights derived from (tensor name, shape, seed) only: the same
call reproduces the same state_dict here, on the GPU box and inside the golden
generator that drives the reference modules (oracle/gen_golden.py).  Scales are
variance-preserving so activations stay in a numerically meaningful range
through 32 residual blocks; this is not a training initialiser.
"""
import math
import zlib

import torch

# buffers that are computed by the constructors (not learned) keep their values
_KEEP = ("div_term",)


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def seeded_tensor(name, shape, seed=1, dtype=torch.float32):
    shape = tuple(shape)
    g = _gen(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return (0.5 + torch.rand(shape, generator=g)).to(dtype)
    if leaf == "running_mean":
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if len(shape) >= 2 and leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return (torch.randn(shape, generator=g) / math.sqrt(3.0 * fan_in)).to(dtype)  # = nn.Linear default variance
    if leaf == "weight":  # 1-D: normalisation gains
        return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "scale":  # LinearAttention.scale, zero-initialised in the reference
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "positional_encoding_gaussian_matrix":  # SAM's random Fourier features: N(0,1) like the constructor
        return torch.randn(shape, generator=g).to(dtype)
    if leaf == "gamma":  # LayerScale gains (DINOv2): positive, O(0.3) like the trained checkpoints' deeper blocks
        return (0.3 + 0.05 * torch.randn(shape, generator=g)).to(dtype)
    if leaf in ("rel_pos_h", "rel_pos_w"):
        return (0.2 * torch.randn(shape, generator=g)).to(dtype)
    # biases, tokens, positional embeddings
    return (0.02 * torch.randn(shape, generator=g)).to(dtype)


def seeded_state(shapes, seed=1, keep=None):
    """shapes: {name: shape}.  keep: optional {name: tensor} for constructor-computed
    buffers (names ending in one of ``_KEEP``) which are passed through untouched."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith(_KEEP):
            if keep is not None and name in keep:
                out[name] = keep[name].clone()
            continue
        out[name] = seeded_tensor(name, shape, seed)
    return out


def load_seeded(module, seed=1):
    """Overwrite every parameter/buffer of ``module`` with its seeded value (in place)."""
    sd = module.state_dict()
    new = seeded_state({k: v.shape for k, v in sd.items()}, seed, keep=sd)
    for k, v in sd.items():
        if k not in new:
            new[k] = v
        else:
            new[k] = new[k].to(v.dtype)
    module.load_state_dict(new, strict=True)
    return module
