"""IFRNet checkpoints (vfi_models/ifrnet/__init__.py:9, IFRNet_L_arch.py / IFRNet_S_arch.py): key/shape table.

Both sizes are the same network with different widths: a 4-level encoder (two PReLU convs per level, the first with
stride 2; the L model's very first conv is 7x7, everything else 3x3) and four decoders
``convrelu -> ResBlock(side) -> ConvTranspose2d(4, 2, 1)``.
"""
CKPT_NAMES = ["IFRNet_S_Vimeo90K.pth", "IFRNet_L_Vimeo90K.pth", "IFRNet_S_GoPro.pth", "IFRNet_L_GoPro.pth"]

# kind -> (encoder widths c1..c4, first kernel size, ResBlock side channels)
CONFIG = {"L": ((64, 96, 144, 192), 7, 64), "S": ((24, 36, 54, 72), 3, 24)}


def kind_of(ckpt_name):
    """The reference picks the small model when the file name contains an 'S' (ifrnet/__init__.py:43)."""
    return "S" if "S" in ckpt_name else "L"


def decoder_io(kind):
    """[(decoder index, Cin of its first conv, trunk width, Cout of its ConvTranspose2d)] for decoder4..decoder1."""
    c1, c2, c3, c4 = CONFIG[kind][0]
    return [(4, 2 * c4 + 1, 2 * c4, 4 + c3), (3, 3 * c3 + 4, 3 * c3, 4 + c2), (2, 3 * c2 + 4, 3 * c2, 4 + c1), (1, 3 * c1 + 4, 3 * c1, 8)]


def ifrnet_shapes(kind):
    """state_dict keys in the reference module's order -> shapes."""
    widths, k0, side = CONFIG[kind]
    sh = {}

    def convrelu(p, cin, cout, k=3):
        sh[p + ".0.weight"] = (cout, cin, k, k)
        sh[p + ".0.bias"] = (cout,)
        sh[p + ".1.weight"] = (cout,)

    cin = 3
    for lvl, c in enumerate(widths, 1):
        convrelu(f"encoder.pyramid{lvl}.0", cin, c, k0 if lvl == 1 else 3)
        convrelu(f"encoder.pyramid{lvl}.1", c, c)
        cin = c
    for d, din, c, dout in decoder_io(kind):
        p = f"decoder{d}.convblock"
        convrelu(p + ".0", din, c)
        for j, cc in ((1, c), (2, side), (3, c), (4, side)):
            convrelu(f"{p}.1.conv{j}", cc, cc)
        sh[p + ".1.conv5.weight"] = (c, c, 3, 3)
        sh[p + ".1.conv5.bias"] = (c,)
        sh[p + ".1.prelu.weight"] = (c,)
        sh[p + ".2.weight"] = (c, dout, 4, 4)
        sh[p + ".2.bias"] = (dout,)
    return sh


def check_state_dict(sd, kind):
    want = ifrnet_shapes(kind)
    missing = [k for k in want if k not in sd]
    extra = [k for k in sd if k not in want]
    if missing or extra:
        raise KeyError(f"IFRNet_{kind} checkpoint: missing keys {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                       f"unexpected keys {extra[:4]}{'...' if len(extra) > 4 else ''}")
    for k, shp in want.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"IFRNet_{kind} checkpoint: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")
