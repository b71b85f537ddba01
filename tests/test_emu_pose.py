"""Pose-solver and point-transformer kernels executed on the HOST through the emulated HIP runtime: the bodies of
tests/test_gpu_pose.py at their small parametrisations (MFMA f32 / split-bf16 MFMA kernels included)."""
import pytest

from tests import test_gpu_pose as T


def test_rot_from_h_on_the_emulator(emu):
    T.test_rot_from_h_vs_svd(emu)


def test_weighted_procrustes_on_the_emulator(emu):
    T.test_weighted_procrustes_vs_oracle_and_batch_invariance(emu, 2, 197)
    T.test_weighted_procrustes_vs_oracle_and_batch_invariance(emu, 1, 7)


def test_half_stored_geo_embedding_on_the_emulator(emu):
    T.test_half_stored_geo_embedding_and_its_reader(emu)


def test_min_dist_on_the_emulator(emu):
    T.test_min_dist_vs_oracle(emu, 196, 3, 1024)
    T.test_min_dist_vs_oracle(emu, 33, 2, 5)                            # a model cloud that is not a multiple of four points


def test_rpe_attention_on_the_emulator(emu):
    T.test_rpe_attention_vs_oracle(emu, 3, 50, "1")
    T.test_rpe_attention_vs_oracle(emu, 3, 50, "0")


def test_geo_embedding_on_the_emulator(emu):
    T.test_geo_embedding_vs_oracle(emu, 1, 37)


@pytest.mark.parametrize("B,M", [(3, 300), (1, 17)])
def test_fine_assign_on_the_emulator(emu, B, M):
    T.test_fine_assign_vs_oracle(emu, B, M)


@pytest.mark.parametrize("B,M1,M2", [(1, 65, 65), (2, 40, 34), (1, 300, 270)])
def test_fine_match_on_the_emulator(emu, B, M1, M2):
    """Fused similarity + assignment (LDS-DMA staged split-bf16 tiles, three sweeps): single tile, ragged sides, several owner
    blocks / odd and even tile counts."""
    T._check_fine_match(emu, B, M1, M2)


def test_coarse_sampling_kernels_on_the_emulator(emu):
    T.test_coarse_sample_vs_oracle(emu, 2, 41, 900)
    T.test_coarse_sample_vs_oracle(emu, 2, 9, 300)          # small M1: the scratch is longer than the bins
    T.test_coarse_sample_vs_oracle(emu, 3, 21, 600)
    T.test_smallest_k_and_hypothesis_select_vs_library(emu)
    T.test_coarse_Rt_kernel_chain_vs_oracle(emu, 2, 40, 300, 30)


def test_positional_encoding_on_the_emulator(emu):
    """Same comparison as T.test_positional_encoding_fused_vs_oracle on a smaller cloud (the emulator is ~1e5 x slower than
    the GPU; the full-size body runs with S6D_EMU_SLOW=1)."""
    import os

    import torch

    from oracle import pem as opem
    from sam6d_amd.pem.pose_estimation_model import PositionalEncoding
    from sam6d_amd.utils import seeded, synth
    if os.environ.get("S6D_EMU_SLOW") == "1":
        return T.test_positional_encoding_fused_vs_oracle(emu)
    pe = seeded.load_seeded(PositionalEncoding(256).eval(), 6)
    W = {"PE." + k: v for k, v in pe.state_dict().items()}
    inp = synth.pem_inputs(1, seed=9, with_rgb=False)
    pts = inp["dense_po"][:, :160].contiguous()
    pts = pts / (pts.norm(dim=2).max(1)[0].reshape(-1, 1, 1) + 1e-6)
    with torch.no_grad():
        ref = opem.positional_encoding(W, "PE", pts)
        assert emu.have("pe_group")
        out = pe(pts)
    assert (out - ref).abs().max() < 5e-5, (out - ref).abs().max()


def test_pose_hypotheses_on_the_emulator(emu):
    T.test_pose_hypotheses_vs_oracle(emu)


def test_cross_and_linear_attention_on_the_emulator(emu):
    """The fused MHA-rows and focused-feature-map kernels inside the product layers vs the oracle layers, at reduced lengths
    (full-size body with S6D_EMU_SLOW=1)."""
    import os

    import torch

    from oracle import pem as opem
    from sam6d_amd.pem.layers import LinearTransformerLayer, TransformerLayer
    from sam6d_amd.utils import seeded
    if os.environ.get("S6D_EMU_SLOW") == "1":
        return T.test_cross_attention_and_linear_attention_vs_oracle(emu)
    g = torch.Generator().manual_seed(8)
    x, mem = torch.randn(2, 37, 256, generator=g), torch.randn(2, 50, 256, generator=g)
    tl = seeded.load_seeded(TransformerLayer(256).eval(), 5)
    W = {"t." + k: v for k, v in tl.state_dict().items()}
    with torch.no_grad():
        ref = opem.cross_layer(W, "t", x, mem)
        assert emu.have("mha") and emu.have("linear_attn_focus")
        out = tl(x, mem)
    assert (out - ref).abs().max() < 2e-5
    ll = seeded.load_seeded(LinearTransformerLayer(256).eval(), 6)
    W = {"l." + k: v for k, v in ll.state_dict().items()}
    xd, ms = torch.randn(1, 200, 256, generator=g), torch.randn(1, 60, 256, generator=g)
    with torch.no_grad():
        ref = opem.linear_layer(W, "l", xd, ms)
        out = ll(xd, ms)
    assert (out - ref).abs().max() < 2e-5


_RPE_VARIANT = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from tests import hipemu
so = hipemu.build(files=["s6d_rpe.hip", "s6d_capi.hip"]) if os.environ.get("HIPEMU_EXTRA") else hipemu.build()
L = ctypes.CDLL(so)
rng = np.random.default_rng(5)
B, N, C = 2, 21, 256                       # 21 keys: the four- and two-key loops end on a partial trip
q, k, v = (rng.standard_normal((B, N, C)).astype(np.float32) for _ in range(3))
qt = (0.1 * rng.standard_normal((B, 4, N, C))).astype(np.float32)
qb = rng.standard_normal((B, 4, N)).astype(np.float32)
emb = rng.standard_normal((B, N, N, C)).astype(np.float32)
out = np.zeros((B, N, C), np.float32)
rc = L.s6d_rpe_attention_f32(hipemu.ptr(q), hipemu.ptr(k), hipemu.ptr(v), hipemu.ptr(qt), hipemu.ptr(qb), hipemu.ptr(emb), B, N, C, 4,
                             ctypes.c_float(0.125), hipemu.ptr(out), None)
assert rc == 0, rc
s = np.einsum("bnhc,bmhc->bhnm", q.reshape(B, N, 4, 64), k.reshape(B, N, 4, 64)) + np.einsum("bhnc,bnmc->bhnm", qt, emb) + qb[..., None]
s = (s * 0.125).astype(np.float64)
p = np.exp(s - s.max(-1, keepdims=True))
p /= p.sum(-1, keepdims=True)
ref = np.einsum("bhnm,bmhc->bnhc", p, v.reshape(B, N, 4, 64)).reshape(B, N, C)
assert np.abs(out - ref).max() < 1e-4, np.abs(out - ref).max()
sys.stdout.buffer.write(out.tobytes())
'''


def test_rpe_keys_per_trip_instantiations_agree_bit_for_bit():
    """csrc/s6d_rpe.hip picks 4 / 1 keys per trip by the number of query rows (and 2 in probe builds); a frame alone and the frame
    inside a launch group must get the same bits.  The three instantiations on the same small input, each in its own host build
    (HIPEMU_EXTRA is read when tests.hipemu is imported) -- the GPU twin at the real threshold is
    tests/test_gpu_pose.py::test_rpe_attention_rows_do_not_depend_on_the_batch_size."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ("", "-DS6D_RPE_FOUR_KEYS_MAX_ROWS=0", "-DS6D_RPE_FOUR_KEYS_MAX_ROWS=0 -DS6D_RPE_KEYS_FULL=2"):
        r = subprocess.run([sys.executable, "-c", _RPE_VARIANT % {"root": root}], env=dict(os.environ, HIPEMU_EXTRA=extra),
                           capture_output=True, timeout=1500)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        outs.append(r.stdout)
    assert len(outs[0]) == 2 * 21 * 256 * 4
    assert outs[0] == outs[1] and outs[0] == outs[2]
