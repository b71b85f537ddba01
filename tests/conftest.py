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


@pytest.fixture(autouse=True)
def _policy_follows_the_environment(monkeypatch):
    """sam6d_amd.policy reads the S6D_* variables ONCE (at import); the forward paths read the policy object.  Tests configure
    through monkeypatch.setenv / delenv: every such call on an S6D_* name re-reads the environment into the policy, and every test
    starts from the environment's policy (the previous test's monkeypatch is undone by then)."""
    from sam6d_amd import policy
    policy.reload()
    policy.reset_library_branch_hits()
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set_(name, value, *a, **k)
        if name.startswith("S6D_"):
            policy.reload()

    def delenv(name, *a, **k):
        del_(name, *a, **k)
        if name.startswith("S6D_"):
            policy.reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield


@pytest.fixture
def emu(monkeypatch):
    """Route sam6d_amd.ops to the HOST build of the kernel sources (tests/hipemu.py: emulated HIP runtime, lanes as fibers,
    MFMA / wave collectives emulated) so that the bodies of the device parity tests can run on CPU tensors at small sizes.
    Test infrastructure only: the product never sees this library."""
    import ctypes

    import torch

    from sam6d_amd import _lib, ops
    from tests import hipemu
    L = ctypes.CDLL(hipemu.build())
    L.s6d_strerror.restype = ctypes.c_char_p                         # as sam6d_amd._lib.lib() sets them on the real library
    L.s6d_strerror.argtypes = [ctypes.c_int]
    L.s6d_last_hip_error.restype = ctypes.c_char_p
    monkeypatch.setattr(_lib, "_lib", L)
    monkeypatch.setattr(ops, "_FUSED", {})
    monkeypatch.setattr(ops, "_stream", lambda: ctypes.c_void_p(0))
    # product modules pick their fused kernels with `x.is_cuda and ops.have(...)`: host tensors claim to be device tensors
    # while the fixture is active, so the same dispatch reaches the emulated kernels
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return ops
