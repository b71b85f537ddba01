import sys, torch
sys.path.insert(0, '.')
from oracle import sam as osam
from sam6d_amd.utils import seeded
from sam6d_amd.sam.image_encoder import build_vit_h
for index in (0, 7):
    m = seeded.load_seeded(build_vit_h().eval(), 3)
    blk = m.blocks[index]
    g = torch.Generator().manual_seed(40 + index)
    x = (0.5 * torch.randn(1, 64, 64, 1280, generator=g)).to(torch.bfloat16)
    name = f"blocks.{index}"
    W = {}
    for k, v in blk.state_dict().items():
        v = v.float()
        if k.endswith(("qkv.weight", "proj.weight", "lin1.weight", "lin2.weight", "rel_pos_h", "rel_pos_w")):
            v = v.to(torch.bfloat16).float()
        W[f"{name}.{k}"] = v
    with torch.no_grad():
        ref = osam.block(W, name, x.float(), 16, blk.window_size)
        m.blocks = torch.nn.ModuleList([blk]); m = m.cuda()
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            out = m._blocks_fused(x.cuda(), None).float().cpu()
    err = (out - ref).abs(); rms = ref.pow(2).mean().sqrt()
    print(index, "rms ratio", float(err.pow(2).mean().sqrt() / rms), "max ratio", float(err.max() / ref.abs().max()), "ref rms", float(rms), "ref max", float(ref.abs().max()))
