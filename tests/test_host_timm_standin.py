"""oracle/timm_standin.py against an INDEPENDENT implementation shipped in the image (VERDICT r5 missing #5).  timm itself cannot be
installed here (the PEM's ViT-B row stays "parity unpinned" at that boundary), but the arithmetic of its pre-LN block --
x + MHA(LN(x)), then x + MLP(LN(x)), fused qkv in [q | k | v] order, exact GELU, softmax(q k^T / sqrt(hd)) v -- is also what
torch.nn.TransformerEncoderLayer(norm_first=True, activation="gelu") and F.scaled_dot_product_attention compute: with the same
weights the stand-in must agree with them to float32 rounding, so the oracle's block is not only checked against itself.
Reference call site: Pose_Estimation_Model/model/feature_extraction.py:7,17-35 (timm VisionTransformer subclass)."""
import torch

from oracle import timm_standin as ts


def _seeded(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.2))
    return m


def test_block_equals_torch_transformer_encoder_layer():
    D, H = 768, 12
    blk = _seeded(ts._Block(D, H, 4.0, True, lambda d: torch.nn.LayerNorm(d, eps=1e-6)).eval(), 3)
    ref = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=4 * D, dropout=0.0, activation="gelu",
                                           layer_norm_eps=1e-6, batch_first=True, norm_first=True).eval()
    with torch.no_grad():
        ref.self_attn.in_proj_weight.copy_(blk.attn.qkv.weight)
        ref.self_attn.in_proj_bias.copy_(blk.attn.qkv.bias)
        ref.self_attn.out_proj.weight.copy_(blk.attn.proj.weight)
        ref.self_attn.out_proj.bias.copy_(blk.attn.proj.bias)
        ref.linear1.weight.copy_(blk.mlp.fc1.weight)
        ref.linear1.bias.copy_(blk.mlp.fc1.bias)
        ref.linear2.weight.copy_(blk.mlp.fc2.weight)
        ref.linear2.bias.copy_(blk.mlp.fc2.bias)
        for a, b in ((ref.norm1, blk.norm1), (ref.norm2, blk.norm2)):
            a.weight.copy_(b.weight)
            a.bias.copy_(b.bias)
    x = torch.randn(2, 197, D, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        # the slow path of the encoder layer (the fused "fast path" kernel is another implementation still; both are checked)
        got = blk(x)
        want_fast = ref(x)
        want = x + ref._sa_block(ref.norm1(x), None, None)
        want = want + ref._ff_block(ref.norm2(want))
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * scale, (got - want).abs().max().item()
    assert (got - want_fast).abs().max().item() <= 2e-5 * scale, (got - want_fast).abs().max().item()


def test_attention_equals_scaled_dot_product_attention():
    D, H = 768, 12
    att = _seeded(ts._Attention(D, H, True).eval(), 7)
    x = torch.randn(3, 197, D, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        got = att(x)
        q, k, v = att.qkv(x).reshape(3, 197, 3, H, D // H).permute(2, 0, 3, 1, 4).unbind(0)
        want = att.proj(torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(3, 197, D))
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()


def test_whole_stand_in_vit_equals_a_stack_of_encoder_layers():
    """Patch embedding as a strided convolution, cls token first, learned positions added after the concatenation, the final norm:
    a two-block ViT of the stand-in against the same statements spelled out with torch's own layers."""
    D, H = 192, 3
    vit = _seeded(ts.VisionTransformer(img_size=64, patch_size=16, embed_dim=D, depth=2, num_heads=H).eval(), 11)
    img = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        x = vit.norm(vit.blocks(vit.norm_pre(vit._pos_embed(vit.patch_embed(img)))))
        # independent spelling
        p = torch.nn.functional.conv2d(img, vit.patch_embed.proj.weight, vit.patch_embed.proj.bias, stride=16).flatten(2).transpose(1, 2)
        y = torch.cat([vit.cls_token.expand(2, -1, -1), p], 1) + vit.pos_embed
        for blk in vit.blocks:
            layer = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=4 * D, dropout=0.0, activation="gelu",
                                                     layer_norm_eps=1e-6, batch_first=True, norm_first=True).eval()
            layer.self_attn.in_proj_weight.copy_(blk.attn.qkv.weight)
            layer.self_attn.in_proj_bias.copy_(blk.attn.qkv.bias)
            layer.self_attn.out_proj.load_state_dict(blk.attn.proj.state_dict())
            layer.linear1.load_state_dict(blk.mlp.fc1.state_dict())
            layer.linear2.load_state_dict(blk.mlp.fc2.state_dict())
            layer.norm1.load_state_dict(blk.norm1.state_dict())
            layer.norm2.load_state_dict(blk.norm2.state_dict())
            y = layer(y)
        y = torch.nn.functional.layer_norm(y, (D,), vit.norm.weight, vit.norm.bias, 1e-6)
    assert (x - y).abs().max().item() <= 3e-5 * y.abs().max().item()
