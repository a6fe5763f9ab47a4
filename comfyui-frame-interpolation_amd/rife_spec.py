"""Checkpoint layout of the RIFE "4.7" architecture family (rife47.pth / rife49.pth).

This is the contract between a reference checkpoint (a plain ``state_dict``
pickle, loaded by ``torch.load`` in vfi_models/rife/__init__.py:131-132) and the
HIP library: the tensors are handed to ``vfi_rife_create`` in exactly the order
returned by :func:`rife47_keys`, each in the reference's own memory layout
(Conv2d ``[Cout,Cin,kh,kw]``, ConvTranspose2d ``[Cin,Cout,kh,kw]``).

Shapes follow vfi_models/rife/rife_arch.py:409-416 (IFNet.__init__, arch "4.7")
and :176-218 (IFBlock.__init__); the key order is torch's ``state_dict`` order
for those modules (own parameters before children, so ``beta`` precedes
``conv.weight``).
"""
from collections import OrderedDict

# rife/__init__.py:10-20 — checkpoint file name -> architecture version.
CKPT_NAME_VER_DICT = {
    "rife47.pth": "4.7",
    "rife49.pth": "4.7",
    "rife417.pth": "4.17",
    "rife426.pth": "4.26",
    "sudo_rife4_269.662_testV1_scale1.pth": "4.0",
}
# architectures the HIP path implements so far
SUPPORTED_ARCH = ("4.7", "4.17", "4.26")   # one fused C-ABI object (vfi_rife_*); "4.0" runs op by op (rife40.py)
# architecture -> code handed to vfi_rife_create
ARCH_CODE = {"4.7": 47, "4.17": 417, "4.26": 426}

# (in_planes, c) per IFBlock, rife_arch.py:410-413
RIFE47_BLOCKS = ((7 + 8, 192), (8 + 4 + 8, 128), (8 + 4 + 8, 96), (8 + 4 + 8, 64))
N_RESCONV = 8
LASTCONV_OUT = 4 * 6  # ConvTranspose2d(c, 4*6, 4, 2, 1) + PixelShuffle(2), rife_arch.py:215-218


def _block_shapes(d, blocks, last_out=None):
    last_out = last_out or LASTCONV_OUT
    for b, (cin, c) in enumerate(blocks):
        p = f"block{b}."
        d[p + "conv0.0.0.weight"] = (c // 2, cin, 3, 3)
        d[p + "conv0.0.0.bias"] = (c // 2,)
        d[p + "conv0.1.0.weight"] = (c, c // 2, 3, 3)
        d[p + "conv0.1.0.bias"] = (c,)
        for i in range(N_RESCONV):
            q = p + f"convblock.{i}."
            d[q + "beta"] = (1, c, 1, 1)
            d[q + "conv.weight"] = (c, c, 3, 3)
            d[q + "conv.bias"] = (c,)
        d[p + "lastconv.0.weight"] = (c, last_out, 4, 4)
        d[p + "lastconv.0.bias"] = (last_out,)


def rife47_shapes():
    """OrderedDict key -> shape, in reference ``state_dict`` order (124 tensors)."""
    d = OrderedDict()
    _block_shapes(d, RIFE47_BLOCKS)
    d["encode.0.weight"] = (16, 3, 3, 3)
    d["encode.0.bias"] = (16,)
    d["encode.1.weight"] = (16, 4, 4, 4)
    d["encode.1.bias"] = (4,)
    return d


# rife_arch.py:417-421 (IFNet.__init__, arch "4.17") + Head_417 :355-375
RIFE417_BLOCKS = ((7 + 16, 192), (8 + 4 + 16, 128), (8 + 4 + 16, 96), (8 + 4 + 16, 64))


def rife417_shapes():
    """rife417.pth: same IFBlocks with 8 feature channels per frame, encoder = Head_417 (128 tensors)."""
    d = OrderedDict()
    _block_shapes(d, RIFE417_BLOCKS)
    for i, (co, ci) in enumerate(((32, 3), (32, 32), (32, 32))):
        d[f"encode.cnn{i}.weight"] = (co, ci, 3, 3)
        d[f"encode.cnn{i}.bias"] = (co,)
    d["encode.cnn3.weight"] = (32, 8, 4, 4)
    d["encode.cnn3.bias"] = (8,)
    return d


# rife_arch.py:451-457 (IFNet.__init__, arch "4.26"): 5 blocks, the later ones also take the 8 feature channels the
# previous block returned; lastconv = ConvTranspose2d(c, 4*13, 4, 2, 1) + PixelShuffle(2) (:221-235); encoder = Head (:378-398)
RIFE426_BLOCKS = ((7 + 8, 192), (8 + 4 + 8 + 8, 128), (8 + 4 + 8 + 8, 96), (8 + 4 + 8 + 8, 64), (8 + 4 + 8 + 8, 32))


def rife426_shapes():
    d = OrderedDict()
    _block_shapes(d, RIFE426_BLOCKS, 4 * 13)
    for i, (co, ci) in enumerate(((16, 3), (16, 16), (16, 16))):
        d[f"encode.cnn{i}.weight"] = (co, ci, 3, 3)
        d[f"encode.cnn{i}.bias"] = (co,)
    d["encode.cnn3.weight"] = (16, 4, 4, 4)
    d["encode.cnn3.bias"] = (4,)
    return d


# rife_arch.py:404-408,460-462 (IFNet.__init__, arch "4.0"): PReLU convs, plain 8-conv trunk, lastconv = ConvTranspose2d(c,5),
# Contextnet (:278-313) and Unet (:316-342)
RIFE40_BLOCKS = ((7, 192), (8 + 4, 128), (8 + 4, 96), (8 + 4, 64))


def rife40_shapes():
    d = OrderedDict()

    def cp(p, cout, cin):
        d[p + ".0.weight"] = (cout, cin, 3, 3)
        d[p + ".0.bias"] = (cout,)
        d[p + ".1.weight"] = (cout,)

    for b, (cin, c) in enumerate(RIFE40_BLOCKS):
        p = f"block{b}."
        cp(p + "conv0.0", c // 2, cin)
        cp(p + "conv0.1", c, c // 2)
        for i in range(N_RESCONV):
            cp(p + f"convblock.{i}", c, c)
        d[p + "lastconv.weight"] = (c, 5, 4, 4)
        d[p + "lastconv.bias"] = (5,)
    for i, (ci, co) in enumerate(((3, 16), (16, 32), (32, 64), (64, 128))):
        cp(f"contextnet.conv{i + 1}.conv1", co, ci)
        cp(f"contextnet.conv{i + 1}.conv2", co, co)
    for i, (ci, co) in enumerate(((17, 32), (64, 64), (128, 128), (256, 256))):
        cp(f"unet.down{i}.conv1", co, ci)
        cp(f"unet.down{i}.conv2", co, co)
    for i, (ci, co) in enumerate(((512, 128), (256, 64), (128, 32), (64, 16))):
        d[f"unet.up{i}.0.weight"] = (ci, co, 4, 4)
        d[f"unet.up{i}.0.bias"] = (co,)
        d[f"unet.up{i}.1.weight"] = (co,)
    d["unet.conv.weight"] = (3, 16, 3, 3)
    d["unet.conv.bias"] = (3,)
    return d


def rife_shapes(arch_ver="4.7"):
    if arch_ver == "4.0":
        return rife40_shapes()
    return {"4.7": rife47_shapes, "4.17": rife417_shapes, "4.26": rife426_shapes}[arch_ver]()


def rife47_keys():
    return list(rife47_shapes().keys())


def check_state_dict(sd, arch_ver="4.7"):
    """Strict key/shape check, same failure mode as ``load_state_dict(strict=True)``."""
    want = rife_shapes(arch_ver)
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    if missing or unexpected:
        raise RuntimeError(
            f"Error(s) in loading state_dict for IFNet({arch_ver}): "
            f"Missing key(s): {missing}. Unexpected key(s): {unexpected}.")
    for k, shp in want.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise RuntimeError(
                f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shp)}")
