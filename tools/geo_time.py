"""geo_embed_kernel timing at the PEM shape (B = 32 clouds of 197 points).  With libraries under tools/geo_variants/ (built from
csrc/s6d_geo.hip + s6d_capi.hip with -D switches, e.g. libgeo_noswz.so = -DS6D_GEO_KSWZ=0) every one is timed in this process and
its output compared with the product library's."""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sam6d_amd import _lib  # noqa: E402

vp = ctypes.c_void_p
g = torch.Generator().manual_seed(0)
B, N = 32, 197
idx4 = (torch.rand(B, N, N, 4, generator=g) * 10).cuda()
Wd, Wa = (torch.randn(256, 256, generator=g) / 16).cuda(), (torch.randn(256, 256, generator=g) / 16).cuda()
bd, ba = torch.randn(256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
div = torch.exp(torch.arange(0, 256, 2).float() * (-9.210340371976184 / 256)).cuda()
NP = idx4.numel() // 4
libs = [("product", _lib.lib())] + [(os.path.basename(p)[6:-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "geo_variants", "libgeo_*.so")))]
ref = None
for rnd in range(2):
    for name, L in libs:
        out = torch.empty(B, N, N, 256, device="cuda")

        def run():
            rc = L.s6d_geo_embedding_f32(vp(idx4.data_ptr()), ctypes.c_long(NP), vp(Wd.data_ptr()), vp(bd.data_ptr()), vp(Wa.data_ptr()),
                                         vp(ba.data_ptr()), vp(div.data_ptr()), 256, 3, vp(out.data_ptr()),
                                         vp(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        print(f"geo_embed {name:10s} {e0.elapsed_time(e1) / 10:.4f} ms   equals product: {bool(torch.equal(out, ref))}", flush=True)
