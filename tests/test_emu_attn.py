"""The MFMA attention kernels and the fused residual + LayerNorm, executed on the HOST through the emulated HIP runtime
(tests/hipemu.py) at small shapes: the bodies of the device parity tests of tests/test_gpu_attn.py, unchanged."""
import pytest

from tests import test_gpu_attn as T


@pytest.mark.parametrize("B,H,nh,hd,ws", [(1, 16, 1, 64, 7), (1, 16, 1, 64, 0), (1, 14, 1, 80, 14),
                                          # item counts that are multiples of 8 take the two-workgroups-per-item kernel:
                                          (1, 28, 2, 80, 14), (1, 14, 2, 64, 7), (2, 20, 4, 80, 14)])
def test_fused_attention_on_the_emulator(emu, B, H, nh, hd, ws):
    T.test_fused_attention_vs_oracle(B, H, nh, hd, ws)


@pytest.mark.parametrize("B,H,nh,hd,ws", [(1, 28, 2, 80, 14), (1, 16, 1, 64, 0), (1, 14, 2, 64, 7)])
def test_head_major_layout_on_the_emulator(emu, B, H, nh, hd, ws):
    T.test_head_major_layout_equals_token_major(B, H, nh, hd, ws)


@pytest.mark.parametrize("grid", [8, 3])
def test_persistent_window_kernel_walks_several_items(emu, monkeypatch, grid):
    """attn_window16p_kernel with fewer workgroups than items (s6d_set_persistent_grid_limit): the LDS-DMA prefetch of the next item's K / V
    images into the other half of LDS, the Q prefetch in the PV pass, scratch tables inside the buffer being filled; 32 items
    (out-of-image window slots included) on 8 and on 3 workgroups (uneven shares, odd and even item counts per workgroup)."""
    from sam6d_amd import _lib
    assert _lib.lib().s6d_set_persistent_grid_limit(grid) == 0
    try:
        T.test_fused_attention_vs_oracle(2, 20, 4, 80, 14)
        T.test_fused_attention_vs_oracle(1, 28, 2, 64, 14)
    finally:
        _lib.lib().s6d_set_persistent_grid_limit(0)


@pytest.mark.parametrize("rows,C", [(37, 160), (5, 768)])
def test_add_layernorm_on_the_emulator(emu, rows, C):
    T.test_add_layernorm_bf16(rows, C)


@pytest.mark.parametrize("B,N,nh,hd", [(2, 50, 2, 80), (1, 70, 1, 64)])
def test_seq_attention_on_the_emulator(emu, B, N, nh, hd):
    T.test_seq_attention_vs_torch(B, N, nh, hd)


def test_segment_seq_sum_and_centroid_path_on_the_emulator(emu, monkeypatch):
    """s6d_segment_seq_sum_f32 through the C ABI (emulated launch) == numpy's row-order reduction, and the library-op path
    of the PEM pre-processing (S6D_PEM_PRE=library -> ops.segment_seq_sum for the centroid) == the oracle loop at boundary-cutting radii."""
    import numpy as np
    import torch

    from oracle import pem_pre as opre
    from sam6d_amd.pem import preprocess as pre
    from sam6d_amd.utils import synth
    g = torch.Generator().manual_seed(0)
    counts = torch.tensor([0, 1, 5, 511, 512, 513, 3000])
    x = torch.randn(int(counts.sum()), 3, generator=g) * 0.3 + 0.8
    start = torch.cumsum(counts, 0) - counts
    got = emu.segment_seq_sum(x, start, counts).numpy()
    for i, (s0, c) in enumerate(zip(start.tolist(), counts.tolist())):
        want = np.add.reduce(x[s0:s0 + c].numpy(), axis=0) if c else np.zeros(3, np.float32)
        np.testing.assert_array_equal(got[i], want)
    monkeypatch.setenv("S6D_PEM_PRE", "library")                     # the library-op path with the centroid kernel in it
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    radius = np.array([0.12, 0.03, 0.5, 0.12, 0.06, 0.2, 0.07, 0.01])
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), radius,
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], torch.from_numpy(radius),
                              keys=inp["keys"], **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])


def test_global_attention_on_the_64_grid_on_the_emulator(emu):
    """SAM's global blocks: the 64 x 64 grid takes the aligned fast path of attn_global_kernel (MODE 1: one key row per 64-key
    tile, th tables per query strip, tw in registers), which the small grids above never reach;
    since round 2 that path is attn_global64_kernel (LDS-DMA ring, swizzled K chunks, counted waits: run under both DMA completion
    models, HIPEMU_GLDS=early / late)."""
    T.test_fused_attention_vs_oracle(1, 64, 1, 80, 0)
    T.test_fused_attention_vs_oracle(1, 64, 1, 64, 0)        # head dim 64: 9-chunk K rows, 19 DMA pieces per tile


def _run_late(cases):
    """Body of the LDS-DMA kernels under the LATE completion model (a DMA's bytes land only at the covering vmcnt wait): run in its
    own interpreter, HIPEMU_GLDS is read once per process."""
    import ctypes

    import torch

    from tests import hipemu
    from sam6d_amd import _lib, ops
    L = ctypes.CDLL(hipemu.build())
    L.s6d_strerror.restype = ctypes.c_char_p
    L.s6d_strerror.argtypes = [ctypes.c_int]
    L.s6d_last_hip_error.restype = ctypes.c_char_p
    _lib._lib = L
    ops._stream = lambda: ctypes.c_void_p(0)
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    for c in cases:
        T.test_fused_attention_vs_oracle(*c)


def test_window_attention_retry_rounds_on_the_emulator(emu):
    T.test_window_attention_when_scores_outgrow_the_first_key_rows(30.0, 1.0)
    T.test_window_attention_when_scores_outgrow_the_first_key_rows(2.0, 1.0)
    T.test_window_attention_when_scores_outgrow_the_first_key_rows(2.0, 2.0 ** 50)      # non-finite accumulators under an in-range row sum


def test_global_attention_fallback_on_the_emulator(emu):
    """Scores that outgrow the first tile's maximum: exp2 overflow -> the workgroup's second pass (30), large finite P (2)."""
    T.test_global_attention_when_scores_outgrow_the_first_tile(80, 30.0, 1.0)
    T.test_global_attention_when_scores_outgrow_the_first_tile(80, 2.0, 1.0)
    T.test_global_attention_when_scores_outgrow_the_first_tile(80, 2.0, 2.0 ** 50)


def test_lds_dma_attention_kernels_under_the_late_completion_model():
    """attn_global64_kernel (64 x 64 grid) and the persistent window kernel with their DMA bytes arriving as late as the waits
    allow: a ring slot read before its counted wait, or refilled before its last reader passed the barrier, shows up here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {root!r}); from tests import test_emu_attn as t; "
                        "t._run_late([(1, 64, 1, 80, 0), (2, 20, 4, 80, 14)])"], env=dict(os.environ, HIPEMU_GLDS="late"),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_float16_kernels_on_the_emulator(emu):
    """IEEE-half builds (PEM ViT-B): sequence attention and add + LayerNorm."""
    T.test_seq_attention_float16_vs_torch(1, 50, 2, 64)
    T.test_add_layernorm_float16()


def test_pem_vit_fused_half_pipeline_on_the_emulator(emu):
    """The PEM ViT (feature_extraction.ViT) through its fused IEEE-half pipeline (s6d_gemm_f16, s6d_seq_attention_f16,
    s6d_add_layernorm_f16) against its own fp32 module path: taps within 2e-3 relative (half's 2^-11 per stored activation), and
    the fused path is really taken (it was not before round 3: the fp32 cls / pos parameters promoted the tokens to fp32)."""
    import torch

    from sam6d_amd.pem.feature_extraction import ViT
    from sam6d_amd.utils import seeded
    m = seeded.load_seeded(ViT(patch_size=16, embed_dim=256, depth=4, num_heads=4, mlp_ratio=2, img_size=64).eval(), 7)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    calls = []
    real = emu.seq_attention
    emu.seq_attention = lambda *a, **k: (calls.append(a[0].dtype), real(*a, **k))[1]
    try:
        with torch.no_grad():
            ref = m(x)
            with torch.autocast(device_type="cpu", dtype=torch.float16):
                out = m.half()(x.half())
    finally:
        emu.seq_attention = real
    assert calls == [torch.float16] * 4
    for a, b in zip(out, ref):
        rel = ((a.float() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
        assert rel < 2e-3, rel
