"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every function that
include/vfi_hip.h declares (no compute calls here)."""
import os
import re

from cfi_amd import _lib, rife_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "vfi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vfi_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(hip_lib):
    names = header_functions()
    assert len(names) >= 15
    assert sorted(_lib.PROTOTYPES) == names
    for n in names:
        assert getattr(hip_lib, n) is not None


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
