"""Import helper: the package directory is named ``comfyui-frame-interpolation_amd`` (a
hyphen, like the reference's own ``ComfyUI-Frame-Interpolation`` checkout), so it is
loaded by path — the same way ComfyUI loads custom-node directories — and registered as
``cfi_amd``."""
import importlib.util
import os
import sys

PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "comfyui-frame-interpolation_amd")
PKG_NAME = "cfi_amd"


def load_package():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        PKG_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
