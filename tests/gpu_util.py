"""Helpers for the -m gpu parity tests: everything goes through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import torch


def ptr(t):
    return C.c_void_p(t.data_ptr())


def hptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def describe_diff(got, want, name="", chan_last=True):
    """Readable summary of where two tensors differ (for assertion messages)."""
    d = (got - want).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    msg = (f"{name}: shape {tuple(got.shape)} max|d|={d.max().item():.3e} at {tuple(int(i) for i in idx)} "
           f"(got {got[idx].item():.6f} want {want[idx].item():.6f}) mean|d|={d.mean().item():.3e} "
           f"frac>1e-3={(d > 1e-3).float().mean().item():.4f} max|want|={want.abs().max().item():.3f}")
    if chan_last and d.dim() >= 2:
        per_c = d.reshape(-1, d.shape[-1]).max(0).values
        msg += " per-channel max: " + " ".join(f"{v:.1e}" for v in per_c[:32].tolist())
    return msg


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()
