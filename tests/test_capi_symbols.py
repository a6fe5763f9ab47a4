"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every function that
include/vfi_hip.h declares (no compute calls here)."""
import os
import re

from cfi_amd import _lib, rife_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="vfi_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vfi_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(hip_lib):
    product, taps = header_functions(), header_functions("vfi_hip_test.h")
    names = sorted(product + taps)
    assert len(product) >= 15 and taps == ["vfi_conv3x3_naive", "vfi_film_debug_read_flow", "vfi_m2m_debug_read", "vfi_rife_debug_keep", "vfi_rife_debug_read", "vfi_test_conv_algo", "vfi_test_film_schedule", "vfi_test_linspace01", "vfi_test_pack_deconv3x3", "vfi_test_pack_wino3x3", "vfi_test_set_option", "vfi_test_variant_override", "vfi_test_wino_probe_read"]   # test taps live apart
    assert sorted(_lib.PROTOTYPES) == product and sorted(_lib.TEST_PROTOTYPES) == taps
    assert _lib.is_test_build()      # tests/conftest.py: the suite runs on libvfi_hip_test.so
    for n in names:
        assert getattr(hip_lib, n) is not None


def test_product_library_has_no_test_taps(hip_lib):
    """libvfi_hip.so — what the package, bench.py and smoke() load — exports every function of include/vfi_hip.h and NONE of
    include/vfi_hip_test.h: nothing in the product can switch a kernel form or read an internal tensor back.
    dlopen only, no call: the process keeps using the test build."""
    import ctypes

    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in header_functions():
        assert hasattr(lib, n), n
    for n in header_functions("vfi_hip_test.h"):
        assert not hasattr(lib, n), f"{n} is exported by the product library"


def test_product_build_is_the_default_in_a_fresh_process():
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); from pkgload import load_package; load_package(); from cfi_amd import _lib; "
            "lib = _lib.load(); assert not _lib.is_test_build() and not hasattr(lib, 'vfi_test_set_option'); "
            "import pytest\n"
            "try:\n    _lib.test_tap('vfi_test_set_option')\nexcept RuntimeError as e:\n    assert 'test tap' in str(e)\nelse:\n    raise SystemExit(1)\n"
            "print('ok')") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_error_string_is_callable(hip_lib):
    assert isinstance(_lib.last_error(), str)


def test_checkpoint_spec_counts():
    shapes = rife_spec.rife47_shapes()
    assert len(shapes) == 124
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    assert n == 5325012


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without libvfi_hip.so the product path raises (it never routes through torch ops or the oracle)."""
    import pytest

    from cfi_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libvfi_hip.so"))
    monkeypatch.setattr(_lib, "TEST_LIB_PATH", str(tmp_path / "libvfi_hip_test.so"))
    with pytest.raises(RuntimeError, match="not built|not found"):
        _lib.load()


def test_engines_need_a_gpu():
    """On a box without a GPU the engines refuse to start instead of computing on the CPU."""
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cfi_amd import synth
    from cfi_amd.film import FilmEngine
    from cfi_amd.m2m import M2MEngine
    from cfi_amd.rife import RifeEngine
    from cfi_amd.rife40 import Rife40Engine

    for ctor, sd in ((lambda s: RifeEngine(s, "4.7"), synth.rife47_synth_state_dict), (FilmEngine, synth.film_synth_state_dict),
                     (M2MEngine, synth.m2m_synth_state_dict), (Rife40Engine, synth.rife40_synth_state_dict)):
        with pytest.raises(RuntimeError, match="no GPU"):
            ctor(sd(1))


def test_reserved_cus_setting_round_trips_without_a_gpu(hip_lib):
    """vfi_set_reserved_cus is plain process state (read by the persistent kernels' launchers): settable and readable on any box,
    out-of-range values refused with a message."""
    try:
        assert hip_lib.vfi_get_reserved_cus() == 0
        assert hip_lib.vfi_set_reserved_cus(16) == 0 and hip_lib.vfi_get_reserved_cus() == 16
        assert hip_lib.vfi_set_reserved_cus(-1) != 0 and "vfi_set_reserved_cus" in _lib.last_error()
        assert hip_lib.vfi_set_reserved_cus(5000) != 0 and hip_lib.vfi_get_reserved_cus() == 16
    finally:
        assert hip_lib.vfi_set_reserved_cus(0) == 0
