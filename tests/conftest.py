import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pkgload import load_package  # noqa: E402

load_package()  # registers the hyphen-named package directory as ``cfi_amd``

from cfi_amd import _lib as _vfi_lib  # noqa: E402

# The suite runs on libvfi_hip_test.so: the product objects + the test taps of include/vfi_hip_test.h (A/B switches, read-backs).
# The product library itself (libvfi_hip.so, what the package loads everywhere else) is covered by tests/test_capi_symbols.py
# (exports), tests/test_gpu_product_build.py (parity, in a child process), bench.py and __graft_entry__.smoke().
_vfi_lib.use_test_build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest`` on a box without a GPU skips the gpu-marked tests instead of erroring in their fixtures."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture()
def oracle_threads():
    """Host-side oracle runs inside GPU tests: torch's default of one thread per logical CPU (256 on the MI355X box) is several
    times SLOWER than 32 for these convolution sizes (bench.py's cpu_baseline measures the same)."""
    import torch

    before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    yield
    torch.set_num_threads(before)


def apply_test_options(lib):
    """``VFI_TEST_OPTIONS="stage_quad=0,fuse_encode=0"`` — a variable of the TEST HARNESS (read here, never by the library): child
    processes of A/B tests select the other of two correct kernel forms through ``vfi_test_set_option`` (include/vfi_hip_test.h)."""
    for item in filter(None, os.environ.get("VFI_TEST_OPTIONS", "").split(",")):
        name, value = item.split("=")
        rc = lib.vfi_test_set_option(name.encode(), int(value))
        assert rc == 0, f"vfi_test_set_option({name!r}) -> {rc}"
    return lib


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; builds it if the in-tree .so is missing or stale."""
    import __graft_entry__ as ge

    ge.build()
    from cfi_amd import _lib

    return apply_test_options(_lib.load())
