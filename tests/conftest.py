import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference (build container only); auto-skipped elsewhere")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(os.environ.get("S6D_REFERENCE_ROOT", "/root/reference"))
    for it in items:
        if "ref" in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason="reference tree not present"))


@pytest.fixture
def emu(monkeypatch):
    """Route sam6d_amd.ops to the HOST build of the kernel sources (tests/hipemu.py: emulated HIP runtime, lanes as fibers,
    MFMA / wave collectives emulated) so that the bodies of the device parity tests can run on CPU tensors at small sizes.
    Test infrastructure only: the product never sees this library."""
    import ctypes

    import torch

    from sam6d_amd import _lib, ops
    from tests import hipemu
    monkeypatch.setattr(_lib, "_lib", ctypes.CDLL(hipemu.build()))
    monkeypatch.setattr(ops, "_FUSED", {})
    real_chk = ops._chk

    class _AsCuda:                                                   # a tensor view whose is_cuda is True for _chk only
        def __init__(self, t):
            self.t = t

        def __getattr__(self, k):
            return True if k == "is_cuda" else getattr(self.t, k)

    monkeypatch.setattr(ops, "_chk", lambda t, dtype, name, ndim=None: real_chk(_AsCuda(t), dtype, name, ndim))
    monkeypatch.setattr(ops, "_stream", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return ops
