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
