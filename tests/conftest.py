import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pkgload import load_package  # noqa: E402

load_package()  # registers the hyphen-named package directory as ``cfi_amd``


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; builds it if the in-tree .so is missing or stale."""
    import __graft_entry__ as ge

    ge.build()
    from cfi_amd import _lib

    return _lib.load()
