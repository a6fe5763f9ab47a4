"""The product library has no experiment switches in its environment: the only variables it reads are the
supported ones of ``_lib.SUPPORTED_ENV``; kernel A/B switches exist only behind ``vfi_test_set_option`` (include/vfi_hip_test.h);
a stray ``VFI_*`` variable is reported loudly and changes nothing (GPU half: tests/test_gpu_env_hygiene.py)."""
import glob
import os
import re
import subprocess
import warnings

import pytest

from cfi_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "comfyui-frame-interpolation_amd")
POISON = {"VFI_WINO_ABLATE": "4", "VFI_WINO16_ABL": "63", "VFI_WINO_2WAVE": "1", "VFI_STAGE_QUAD": "0", "VFI_SPLAT_MODE": "atomic",
          "VFI_CONV_WINOGRAD": "0", "VFI_RIFE_FUSE0A": "0", "VFI_RIFE_FUSE_ENCODE": "0", "VFI_CONV_SPLITK": "0", "VFI_GROUPED_VARIANT": "12",
          "VFI_WINO_XCD": "0", "VFI_VARIANT_OVERRIDE": "resconv_c64=36", "VFI_CONV_M2N2_PX": "1", "VFI_SPLAT_SPILL_CAP": "1"}


def test_supported_set_is_small_and_documented():
    assert len(_lib.SUPPORTED_ENV) <= 9      # r6: + VFI_PAIR_LANES (a resource knob: frames are bit-identical for any value)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in _lib.SUPPORTED_ENV:
        assert name in doc, f"{name} is not documented in INTEGRATION.md"


def test_sources_read_only_supported_variables():
    read = set()
    for f in glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")):
        read |= set(re.findall(r'getenv\(\s*"([A-Z0-9_]+)"', open(f).read()))
    for f in glob.glob(os.path.join(PKG, "*.py")):
        read |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(VFI_[A-Z0-9_]+)"', open(f).read()))
    assert read and read <= set(_lib.SUPPORTED_ENV), sorted(read - set(_lib.SUPPORTED_ENV))


def test_library_binary_has_no_experiment_switch_names(hip_lib):
    out = subprocess.run(["strings", "-n", "4", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = set(re.findall(r"\bVFI_[A-Z0-9_]{3,}\b", out))
    assert names <= set(_lib.SUPPORTED_ENV), sorted(names - set(_lib.SUPPORTED_ENV))
    assert not re.search(r"ABLATE|WINO16|_ABL\b|wino16", out)


def test_stray_variables_are_reported(monkeypatch):
    for k in list(os.environ):
        if k.startswith("VFI_"):
            monkeypatch.delenv(k)
    assert _lib.audit_environment() == []
    monkeypatch.setenv("VFI_DEVICES", "current")
    monkeypatch.setenv("VFI_TEST_OPTIONS", "")
    assert _lib.audit_environment() == []
    for k, v in POISON.items():
        monkeypatch.setenv(k, v)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _lib.audit_environment() == sorted(POISON)
    assert len(w) == 1 and issubclass(w[0].category, RuntimeWarning) and "VFI_WINO_ABLATE" in str(w[0].message)


def test_unknown_option_is_refused(hip_lib):
    assert hip_lib.vfi_test_set_option(b"no_such_option", 1) == -2 and "unknown option" in _lib.last_error()
    assert hip_lib.vfi_test_set_option(b"wino_xcd", 1) == 0
