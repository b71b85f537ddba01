"""CPU restatement of the SAM image encoder -- TEST INFRASTRUCTURE ONLY.

Functional torch-fp32 restatement of
``Instance_Segmentation_Model/segment_anything/modeling/image_encoder.py`` (ImageEncoderViT,
Block, Attention, window_partition/unpartition, get_rel_pos, add_decomposed_rel_pos,
PatchEmbed), ``modeling/common.py`` (MLPBlock, LayerNorm2d) and ``Sam.preprocess``
(modeling/sam.py:164-174).  Weights: flat {state_dict key: tensor} with the reference key
names under ``image_encoder.`` stripped.  Pinned by tests/golden/sam_*.npz (reference
modules imported unmodified by oracle/gen_golden.py).
"""
import torch
import torch.nn.functional as F

VIT_H = dict(img_size=1024, patch=16, dim=1280, depth=32, heads=16, window=14,
             global_idx=(7, 15, 23, 31), out_chans=256)
MINI = dict(img_size=512, patch=16, dim=160, depth=4, heads=2, window=14,
            global_idx=(1, 3), out_chans=64)


def preprocess(x, img_size=1024, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
    """Sam.preprocess: normalise then zero-pad bottom/right to img_size (sam.py:164-174)."""
    m = torch.tensor(mean).view(-1, 1, 1)
    s = torch.tensor(std).view(-1, 1, 1)
    x = (x - m) / s
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))


def _rel_pos(q_size, k_size, rel_pos):
    """get_rel_pos (image_encoder.py:292-322)."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel, mode="linear")
        rel_pos = r.reshape(-1, max_rel).permute(1, 0)
    qc = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.long()]


def attention_from_qkv(qkv, rel_pos_h, rel_pos_w, heads):
    """The attention statements of Attention.forward between the two Linear layers + add_decomposed_rel_pos
    (image_encoder.py:229-238, 325-361) on a given qkv (B,H,W,3C).  NOTE: the rel-pos terms use the UNscaled q."""
    B, H, Wd, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    qkv = qkv.reshape(B, H * Wd, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * Wd, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    if rel_pos_h is not None:
        Rh = _rel_pos(H, H, rel_pos_h)
        Rw = _rel_pos(Wd, Wd, rel_pos_w)
        rq = q.reshape(B * heads, H, Wd, hd)
        rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
        attn = (attn.view(B * heads, H, Wd, H, Wd) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(
            B * heads, H * Wd, H * Wd)
    attn = attn.softmax(-1)
    return (attn @ v).view(B, heads, H, Wd, hd).permute(0, 2, 3, 1, 4).reshape(B, H, Wd, C)


def attention(W, p, x, heads):
    """Attention.forward (image_encoder.py:224-240).  x (B,H,W,C)."""
    qkv = F.linear(x, W[p + ".qkv.weight"], W[p + ".qkv.bias"])
    return F.linear(attention_from_qkv(qkv, W[p + ".rel_pos_h"], W[p + ".rel_pos_w"], heads), W[p + ".proj.weight"],
                    W[p + ".proj.bias"])


def windowed_attention_from_qkv(qkv, qkv_bias, rel_pos_h, rel_pos_w, heads, window):
    """What Block.forward computes between the qkv Linear and the proj Linear (image_encoder.py:168-179 around Attention), on
    the UN-padded qkv (B,H,W,3C) of the token map: quirk Q2 -- the reference pads AFTER norm1, so a padded token is a zero vector
    whose q / k / v equal the qkv bias exactly (Linear(0) = bias) and it takes part as a key; padding the qkv map with the bias
    is the same statement.  window = 0: global attention.  This is the comparand of the fused attention kernels."""
    if window == 0:
        return attention_from_qkv(qkv, rel_pos_h, rel_pos_w, heads)
    B, H, Wd, C3 = qkv.shape
    ph, pw = (window - H % window) % window, (window - Wd % window) % window
    Hp, Wp = H + ph, Wd + pw
    full = qkv_bias.to(qkv.dtype).expand(B, Hp, Wp, C3).clone()
    full[:, :H, :Wd] = qkv
    x = full.view(B, Hp // window, window, Wp // window, window, C3).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, C3)
    x = attention_from_qkv(x, rel_pos_h, rel_pos_w, heads)
    C = C3 // 3
    x = x.view(B, Hp // window, Wp // window, window, window, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    return x[:, :H, :Wd, :]


def block(W, p, x, heads, window):
    """Block.forward (image_encoder.py:166-182).  Quirk Q2: padding happens AFTER norm1, so the
    padded tokens are zeros whose q/k/v equal the qkv bias and they DO take part as keys (windowed_attention_from_qkv)."""
    sc = x
    x = F.layer_norm(x, (x.shape[-1],), W[p + ".norm1.weight"], W[p + ".norm1.bias"], 1e-6)
    a = p + ".attn"
    qkv = F.linear(x, W[a + ".qkv.weight"], W[a + ".qkv.bias"])
    x = windowed_attention_from_qkv(qkv, W[a + ".qkv.bias"], W[a + ".rel_pos_h"], W[a + ".rel_pos_w"], heads, window)
    x = F.linear(x, W[a + ".proj.weight"], W[a + ".proj.bias"])
    x = sc + x
    h = F.layer_norm(x, (x.shape[-1],), W[p + ".norm2.weight"], W[p + ".norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, W[p + ".mlp.lin1.weight"], W[p + ".mlp.lin1.bias"])),
                 W[p + ".mlp.lin2.weight"], W[p + ".mlp.lin2.bias"])
    return x + h


def _ln2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def encoder_forward(W, x, cfg=VIT_H, upto=None):
    """ImageEncoderViT.forward (image_encoder.py:106-116).  ``upto``: stop after that many
    blocks and return the token map (B,H,W,C) (for per-block parity checks)."""
    x = F.conv2d(x, W["patch_embed.proj.weight"], W["patch_embed.proj.bias"], stride=cfg["patch"]).permute(0, 2, 3, 1)
    x = x + W["pos_embed"]
    for i in range(cfg["depth"]):
        if upto is not None and i >= upto:
            return x
        x = block(W, f"blocks.{i}", x, cfg["heads"], 0 if i in cfg["global_idx"] else cfg["window"])
    if upto is not None:
        return x
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, W["neck.0.weight"])
    x = _ln2d(x, W["neck.1.weight"], W["neck.1.bias"])
    x = F.conv2d(x, W["neck.2.weight"], padding=1)
    return _ln2d(x, W["neck.3.weight"], W["neck.3.bias"])
