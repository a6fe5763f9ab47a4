"""MI355X-native drop-in for the RIFE / FILM / M2M / IFRNet nodes of ComfyUI-Frame-Interpolation.

ComfyUI imports this directory as a custom-node package and reads
``NODE_CLASS_MAPPINGS`` (reference: /root/reference/__init__.py:24-48).  The node
classes are imported lazily so that tooling which only needs the checkpoint spec or the
scheduler does not require the HIP library to be built.
"""

_LAZY = {
    "RIFE_VFI": ("rife", "RIFE_VFI"),
    "FILM_VFI": ("film", "FILM_VFI"),
    "M2M_VFI": ("m2m", "M2M_VFI"),
    "IFRNet_VFI": ("ifrnet", "IFRNet_VFI"),
    "GMFSS_Fortuna_VFI": ("gmfss", "GMFSS_Fortuna_VFI"),
    "IFUnet_VFI": ("ifunet", "IFUnet_VFI"),
    "MakeInterpolationStateList": ("schedule", "MakeInterpolationStateList"),
    "InterpolationStateList": ("schedule", "InterpolationStateList"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    if name == "NODE_CLASS_MAPPINGS":
        return _node_class_mappings()
    raise AttributeError(name)


def _node_class_mappings():
    from .film import FILM_VFI
    from .gmfss import GMFSS_Fortuna_VFI
    from .ifunet import IFUnet_VFI
    from .ifrnet import IFRNet_VFI
    from .m2m import M2M_VFI
    from .rife import RIFE_VFI
    from .schedule import MakeInterpolationStateList

    return {
        "RIFE VFI": RIFE_VFI,
        "FILM VFI": FILM_VFI,
        "M2M VFI": M2M_VFI,
        "IFRNet VFI": IFRNet_VFI,
        "GMFSS Fortuna VFI": GMFSS_Fortuna_VFI,
        "IFUnet VFI": IFUnet_VFI,
        "Make Interpolation State List": MakeInterpolationStateList,
    }


NODE_DISPLAY_NAME_MAPPINGS = {
    "RIFE VFI": "RIFE VFI (MI355X HIP; rife47 / rife49)",
    "FILM VFI": "FILM VFI (MI355X HIP)",
    "M2M VFI": "M2M VFI (MI355X HIP)",
    "IFRNet VFI": "IFRNet VFI (MI355X HIP)",
    "GMFSS Fortuna VFI": "GMFSS Fortuna VFI (MI355X HIP)",
    "IFUnet VFI": "IFUnet VFI (MI355X HIP)",
}
