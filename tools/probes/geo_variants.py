import ctypes, glob, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sam6d_amd import ops  # noqa
from sam6d_amd.pem.layers import GeometricStructureEmbedding
from sam6d_amd.pem.pose_estimation_model import default_cfg
from sam6d_amd.utils import seeded, synth
geo = seeded.load_seeded(GeometricStructureEmbedding(default_cfg().geo_embedding).eval(), 4).cuda()
split = geo._split_weights()
pts = synth.pem_inputs(32, seed=9, n_pts=197, with_rgb=False)["pts"].cuda() * 5
d_idx, a_idx = geo.get_embedding_indices(pts)
idx4 = torch.cat([d_idx.unsqueeze(-1), a_idx], dim=-1).contiguous()
NP = idx4.numel() // 4
out = torch.empty(NP, 256, device="cuda")
bd, ba, dt = geo.proj_d.bias.contiguous(), geo.proj_a.bias.contiguous(), geo.embedding.div_term.contiguous()
res = {}
libs = sorted(glob.glob(os.path.join(sys.argv[1], "libgeo_*.so")))
for rnd in range(2):
    for so in libs:
        L = ctypes.CDLL(so)
        f = L.s6d_geo_embedding_split
        f.restype = ctypes.c_int
        args = (ctypes.c_void_p(idx4.data_ptr()), ctypes.c_long(NP), ctypes.c_void_p(split[0].data_ptr()), ctypes.c_void_p(bd.data_ptr()),
                ctypes.c_void_p(split[1].data_ptr()), ctypes.c_void_p(ba.data_ptr()), ctypes.c_void_p(dt.data_ptr()), 256, 3,
                ctypes.c_void_p(out.data_ptr()), 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3):
            assert f(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            f(*args)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(os.path.basename(so), []).append(round(e0.elapsed_time(e1) / 20, 4))
print(json.dumps(res, indent=1))
