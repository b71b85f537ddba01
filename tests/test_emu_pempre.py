"""The PEM sampler kernel (csrc/s6d_pempre.hip) on the emulator against the library formulation it replaces (the top-k over
64-bit composite keys in sam6d_amd/pem/preprocess.py): identical indices, ties resolved by position, duplicated keys flagged."""
import numpy as np
import torch

from sam6d_amd.pem import preprocess as pre


def _lib_path(n, keys, n_sample, monkeypatch):
    monkeypatch.delenv("S6D_PEM_SAMPLER", raising=False)
    return pre._keyed_indices(n, keys, n_sample)


def test_sampler_kernel_equals_the_library_path(emu, monkeypatch):
    g = torch.Generator().manual_seed(0)
    L, ns = 40000, 512
    n = torch.tensor([0, 1, 100, 512, 513, 600, 1024, 5000, 40000, 2049])
    keys = torch.rand(len(n), L, generator=g)
    keys[5] = (keys[5] * 300).floor() / 300                          # ties among 600 keys: the position decides
    keys[7, :5000] = (keys[7, :5000] * 1e4).floor() / 1e4
    want = _lib_path(n, keys, ns, monkeypatch)
    idx, overflow = emu.pem_sample_indices(keys, n, ns)
    assert overflow.tolist() == [0] * len(n)
    assert torch.equal(idx, want)
    # the selected positions are what a stable argsort gives
    row = 8
    np.testing.assert_array_equal(idx[row].numpy(), np.argsort(keys[row].numpy(), kind="stable")[:ns])
    # through observed_inputs' switch as well
    monkeypatch.setenv("S6D_PEM_SAMPLER", "kernel")
    assert torch.equal(pre._keyed_indices(n, keys, ns), want)


def test_heavily_duplicated_keys_are_flagged_and_fall_back(emu, monkeypatch):
    g = torch.Generator().manual_seed(1)
    n = torch.tensor([30000, 30000])
    keys = torch.rand(2, 30000, generator=g)
    keys[1] = (keys[1] * 4).floor() / 4                              # four distinct values: 7500 keys share the smallest
    want = _lib_path(n, keys, 2048, monkeypatch)
    idx, overflow = emu.pem_sample_indices(keys, n, 2048)
    assert overflow.tolist() == [0, 1] and torch.equal(idx[0], want[0])
    monkeypatch.setenv("S6D_PEM_SAMPLER", "kernel")
    assert torch.equal(pre._keyed_indices(n, keys, 2048), want)     # the switch falls back to the library path for the frame


def test_full_frame_with_the_kernel_sampler_matches_the_oracle(emu, monkeypatch):
    from oracle import pem_pre as opre
    from sam6d_amd.utils import synth
    monkeypatch.setenv("S6D_PEM_SAMPLER", "kernel")
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(radius=0.12, n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(),
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], keys=inp["keys"], **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])
