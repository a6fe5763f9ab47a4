"""Checkpoint layout of M2M (M2M.pth, a plain state_dict of M2M_PWC; vfi_models/m2m/__init__.py:43-45).

Key names / shapes follow vfi_models/m2m/M2M_arch.py (M2M_PWC.__init__ :850-892 and the modules it builds);
order = torch state_dict order (own parameters before children)."""
from collections import OrderedDict

BRANCH = 4
C = 16  # M2M_arch.py:586


def m2m_shapes():
    d = OrderedDict()
    d["paramAlpha"] = (1, 1, 1, 1)
    for name, cin in (("netOne", 3), ("netTwo", 32), ("netThr", 32)):
        p = f"netFlow.netExtractor.{name}.netMain."
        d[p + "0.weight"] = (32, cin, 2, 2)
        d[p + "0.bias"] = (32,)
        d[p + "1.weight"] = (1,)
        for i in (2, 4):
            d[p + f"{i}.weight"] = (32, 32, 3, 3)
            d[p + f"{i}.bias"] = (32,)
            d[p + f"{i + 1}.weight"] = (1,)
    for name, cin in (("netFiv", 113), ("netFou", 115), ("netThr", 115), ("netTwo", 115), ("netOne", 115)):
        d[f"netFlow.{name}.netCostacti.weight"] = (1,)
        chans = [cin, 128, 128, 96, 64, 32, 2]
        p = f"netFlow.{name}.netMain.netMain."
        for i in range(6):
            d[p + f"{2 * i}.weight"] = (chans[i + 1], chans[i], 3, 3)
            d[p + f"{2 * i}.bias"] = (chans[i + 1],)
            if i < 5:
                d[p + f"{2 * i + 1}.weight"] = (1,)

    def conv2(p, cin, cout):
        for name, ci in (("conv1", cin), ("conv2", cout)):
            d[f"{p}.{name}.0.weight"] = (cout, ci, 3, 3)
            d[f"{p}.{name}.0.bias"] = (cout,)
            d[f"{p}.{name}.1.weight"] = (cout,)

    for i, (ci, co) in enumerate(((3, C), (C, 2 * C), (2 * C, 4 * C), (4 * C, 8 * C))):
        conv2(f"MRN.img_pyramid.conv{i + 1}", ci, co)
    for i, (ci, co) in enumerate(((8, 2 * C), (6 * C, 4 * C), (12 * C, 8 * C), (24 * C, 16 * C))):
        conv2(f"MRN.motion_encdec.down{i}", ci, co)
    for i, (ci, co) in enumerate(((48 * C, 8 * C), (16 * C, 4 * C), (8 * C, 2 * C), (4 * C, C))):
        d[f"MRN.motion_encdec.up{i}.0.weight"] = (ci, co, 4, 4)
        d[f"MRN.motion_encdec.up{i}.0.bias"] = (co,)
        d[f"MRN.motion_encdec.up{i}.1.weight"] = (co,)
    d["MRN.motion_encdec.conv.weight"] = (2 * BRANCH, C, 3, 3)
    d["MRN.motion_encdec.conv.bias"] = (2 * BRANCH,)
    d["MRN.motion_encdec.conv_m.weight"] = (1, C, 3, 3)
    d["MRN.motion_encdec.conv_m.bias"] = (1,)
    d["MRN.motion_encdec.conv_C.1.weight"] = (16 * 16 * C, 16 * C, 1, 1)
    d["MRN.motion_encdec.conv_C.1.bias"] = (16 * 16 * C,)
    for n in ("conv_H", "conv_W"):
        d[f"MRN.motion_encdec.{n}.1.weight"] = (16, 16 * C, 1, 1)
        d[f"MRN.motion_encdec.{n}.1.bias"] = (16,)
    return d


def check_state_dict(sd):
    want = m2m_shapes()
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    if missing or unexpected:
        raise RuntimeError(f"Error(s) in loading state_dict for M2M_PWC: Missing key(s): {missing}. Unexpected key(s): {unexpected}.")
    for k, shp in want.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shp)}")
