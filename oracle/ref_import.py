"""ORACLE tooling — import the real reference modules from /root/reference on CPU.

Only usable where /root/reference exists (this build container).  Nothing in the
``-m gpu`` tests, ``smoke()`` or ``bench.py`` touches this module.

Recipe: put oracle/stubs (comfy.model_management, torchvision,
cv2 shims) ahead of /root/reference on sys.path; ``vfi_models`` is a namespace package.
"""
import os
import sys

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REFERENCE, "vfi_models"))


def setup():
    if not available():
        raise RuntimeError("reference checkout not present at /root/reference")
    stubs = os.path.join(HERE, "stubs")
    for p in (REFERENCE, stubs):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, stubs)
    sys.dont_write_bytecode = True  # /root/reference is read-only


def rife_arch():
    setup()
    import vfi_models.rife.rife_arch as m

    return m


def rife_node(state_dict_path):
    """The reference RIFE_VFI class with its downloader redirected to a local .pth."""
    setup()
    import vfi_models.rife as R

    R.load_file_from_github_release = lambda model_type, ckpt: state_dict_path
    R._model_cache.clear()
    return R


def vfi_utils():
    setup()
    import vfi_utils as m

    return m


def reference_ops():
    """The reference's OWN ``vfi_models.ops`` (config.yaml: ops_backend "cupy" -> vfi_models/ops/cupy_ops) running on the
    host: oracle/stubs/cupy compiles the kernel text the reference's ``cuda_kernel`` specialises (cupy_ops/utils.py:29-213)
    with g++ behind a serial shim.  The op wrappers insist on CUDA tensors (softsplat.py:205,226: ``assert False`` for
    CPU tensors), so for the life of this process host tensors report ``is_cuda`` and the two torch.cuda calls on the
    launch path (utils.py:31 get_device_name, softsplat.py:221 current_stream) are answered by constants."""
    import collections

    import torch

    setup()
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.cuda.get_device_name = lambda *a, **k: "host-shim"
    torch.cuda.current_stream = lambda *a, **k: collections.namedtuple("S", "cuda_stream")(0)
    import vfi_models.ops as ops

    return ops
