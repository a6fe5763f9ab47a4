"""Import shim so the reference modules import on a CPU-only box (oracle use only).

The reference pulls exactly these names from ComfyUI (vfi_utils.py:11,
vfi_models/rife/rife_arch.py:14, vfi_models/ops/cupy_ops/utils.py)."""
import torch


def get_torch_device():
    return torch.device("cpu")


def soft_empty_cache(*a, **k):
    return None


def is_nvidia():
    return False


def get_torch_device_name(d):
    return str(d)
