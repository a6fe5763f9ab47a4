"""Checkpoint location / download with the reference's directory contract.

Behaviour mirrored from vfi_utils.py:84-137: a checkpoint lives at
``<package>/<ckpts_path>/<model_type>/<ckpt_name>`` (``ckpts_path`` from config.yaml); if it is
not there every base URL and then the per-file mirrors are tried in order, and one combined
exception is raised when all fail.
"""
import os
import traceback

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))

BASE_MODEL_DOWNLOAD_URLS = [
    "https://github.com/styler00dollar/VSGAN-tensorrt-docker/releases/download/models/",
    "https://github.com/Fannovel16/ComfyUI-Frame-Interpolation/releases/download/models/",
    "https://github.com/dajes/frame-interpolation-pytorch/releases/download/v1.0.0/",
]
CKPT_FALLBACK_URLS = {
    "rife47.pth": [
        "https://huggingface.co/marduk191/rife/resolve/main/rife47.pth",
        "https://huggingface.co/wavespeed/misc/resolve/main/rife/rife47.pth",
        "https://huggingface.co/MachineDelusions/RIFE/resolve/main/rife47.pth",
    ],
    "rife49.pth": [
        "https://huggingface.co/marduk191/rife/resolve/main/rife49.pth",
        "https://huggingface.co/hfmaster/models-moved/resolve/main/rife/rife49.pth",
        "https://huggingface.co/MachineDelusions/RIFE/resolve/main/rife49.pth",
    ],
}


def load_config():
    path = os.path.join(_HERE, "config.yaml")
    if not os.path.exists(path):
        raise Exception("config.yaml file is necessary next to the node package")
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def get_ckpt_container_path(model_type):
    return os.path.abspath(os.path.join(_HERE, load_config()["ckpts_path"], model_type))


def load_file_from_github_release(model_type, ckpt_name):
    model_dir = get_ckpt_container_path(model_type)
    cached = os.path.join(model_dir, ckpt_name)
    if os.path.exists(cached):
        return cached
    from torch.hub import download_url_to_file

    os.makedirs(model_dir, exist_ok=True)
    urls = [base + ckpt_name for base in BASE_MODEL_DOWNLOAD_URLS] + CKPT_FALLBACK_URLS.get(ckpt_name, [])
    errors = []
    for url in urls:
        try:
            print(f'Downloading: "{url}" to {cached}\n')
            download_url_to_file(url, cached, hash_prefix=None, progress=True)
            return cached
        except Exception:
            errors.append(f"Error when downloading from: {url}\n\n{traceback.format_exc()}")
    raise Exception(f"Tried all urls to download {ckpt_name} but no success. Below is the error log:\n\n"
                    + "\n\n".join(errors))


# ---- engine cache -------------------------------------------------------------------------------------------------
# The reference caches its RIFE module across node executions (rife/__init__.py:29-31) but rebuilds FILM and M2M on
# every call.  Building an engine means re-packing and uploading every weight (~0.15 s for FILM's 138 MB), so the HIP
# nodes keep the device-resident weights of the last checkpoint per model type; activations are released after each
# call.  Keyed by path + size + mtime: a replaced file is reloaded.  VFI_MODEL_CACHE=0 restores load-per-call.
_engine_cache = {}


def cached_engine(model_type, path, build):
    """build(): -> engine with .close() and .release_workspace().  Returns (engine, cached: bool)."""
    import os

    if os.environ.get("VFI_MODEL_CACHE", "1") == "0":
        return build(), False
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_size, st.st_mtime_ns)
    hit = _engine_cache.get(model_type)
    if hit is not None and hit[0] == key:
        return hit[1], True
    if hit is not None:
        hit[1].close()
    eng = build()
    _engine_cache[model_type] = (key, eng)
    return eng, True


# A cached engine's WORKSPACE (activations, pooled scratch, captured graphs) normally goes back at the end of a node call: the next call
# re-allocates it (M2M +2 ms, FILM +7 ms / 15 GB).  The op-by-op engines pay far more for it — a lane's first two pairs of a call cost
# GMFSS 150 ms and IFUNet 130 ms over steady ones (pool chunks, graph capture; tools/engine_build_probe.py) — for 2.7-3.1 GiB per lane: on a
# 288 GB device those stay resident between calls of the same frame shape (a new shape, or a set above the budget, releases as before).
KEEP_WORKSPACE_BYTES = 16 << 30


def begin_call(engine, shape):
    """before a cached engine's node call: a kept workspace is for ONE frame shape (its root tensors are keyed by shape and would pile up)"""
    kept = getattr(engine, "_kept_shape", None)
    if kept is not None and kept != tuple(shape):
        engine.release_workspace()
    engine._kept_shape = tuple(shape)


def end_call(engine, cached):
    """after a node call: close an uncached engine; release a cached one's workspace unless it may stay (KEEP_WORKSPACE_BYTES)"""
    if not cached:
        engine.close()
        return
    size = engine.workspace_bytes() if hasattr(engine, "workspace_bytes") else None
    if size is None or size > KEEP_WORKSPACE_BYTES:
        engine.release_workspace()
        engine._kept_shape = None


def clear_engine_cache():
    for _, eng in _engine_cache.values():
        eng.close()
    _engine_cache.clear()
