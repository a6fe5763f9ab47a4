"""IFUNet.pth (vfi_models/ifunet/__init__.py:9, IFUNet_arch.py): key/shape table of IFUNetModel = flownet (IFUNet: CBAM U-Net
FeatureNet + 3 convex-up-sampling IFBlocks) + fusionnet (RRDBNet, 6 blocks) + refinenet (ResynNet: 3 BatchNorm FlowBlocks,
context / decode, and the training-only DegCNN that the checkpoint still carries).  Order = the reference module's."""
from collections import OrderedDict

CKPT_NAMES = ["IFUNet.pth"]


def _conv(d, p, cin, cout, k=3):
    """conv(): Conv2d + PReLU (IFUNet_arch.py:18-30)"""
    d[p + ".0.weight"] = (cout, cin, k, k)
    d[p + ".0.bias"] = (cout,)
    d[p + ".1.weight"] = (cout,)


def _bn(d, p, c):
    for n in ("weight", "bias", "running_mean", "running_var"):
        d[f"{p}.{n}"] = (c,)
    d[p + ".num_batches_tracked"] = ()


def _cbam(d, p, c):
    d[p + ".ChannelGate.mlp.1.weight"] = (c // 16, c)
    d[p + ".ChannelGate.mlp.1.bias"] = (c // 16,)
    d[p + ".ChannelGate.mlp.3.weight"] = (c, c // 16)
    d[p + ".ChannelGate.mlp.3.bias"] = (c,)
    d[p + ".SpatialGate.spatial.conv.weight"] = (1, 2, 7, 7)
    _bn(d, p + ".SpatialGate.spatial.bn", 1)


def ifunet_shapes():
    d = OrderedDict()
    p = "flownet.fmap"
    _conv(d, p + ".conv0", 7, 17, 1)
    cin = 17
    for i, (c, att) in enumerate(((32, False), (64, True), (128, True), (256, True), (512, True)), 1):   # UNetConv :521-537
        _conv(d, f"{p}.conv{i}.conv1", cin, c)
        _conv(d, f"{p}.conv{i}.conv2", c, c)
        if att:
            _cbam(d, f"{p}.conv{i}.cbam", c)
        cin = c
    for i, (c, cout, att) in ((5, (512, 256, True)), (4, (256, 128, False)), (3, (128, 64, False))):   # UpConv :540-563
        q = f"{p}.deconv{i}"
        d[q + ".deconv.0.weight"] = (c, c // 2, 4, 4)
        d[q + ".deconv.0.bias"] = (c // 2,)
        d[q + ".deconv.1.weight"] = (c // 2,)
        _conv(d, q + ".conv1", c, c // 2)
        _conv(d, q + ".conv2", c // 2, cout)
        if att:
            _cbam(d, q + ".cbam", cout)
    for b, c in enumerate((256, 128, 64)):   # IFBlock :600-617
        q = f"flownet.block{b}"
        for i in range(6):
            _conv(d, f"{q}.convblock.{i}", c, c)
        d[q + ".flowconv.weight"] = (4, c, 3, 3)
        d[q + ".flowconv.bias"] = (4,)
        for lvl in (16, 8, 4):
            d[f"{q}.maskconvx{lvl}.weight"] = (lvl * lvl * 9, c, 1, 1)
            d[f"{q}.maskconvx{lvl}.bias"] = (lvl * lvl * 9,)
    q = "fusionnet"   # RRDBNet(num_in_ch=16, num_out_ch=1, num_feat=64, num_block=6, num_grow_ch=32) :269-304
    d[q + ".conv_first.weight"] = (64, 16, 3, 3)
    d[q + ".conv_first.bias"] = (64,)
    for b in range(6):
        for r in (1, 2, 3):
            for k in range(1, 6):
                cout = 32 if k < 5 else 64
                d[f"{q}.body.{b}.rdb{r}.conv{k}.weight"] = (cout, 64 + 32 * (k - 1), 3, 3)
                d[f"{q}.body.{b}.rdb{r}.conv{k}.bias"] = (cout,)
    for name, cout in (("conv_body", 64), ("conv_up1", 64), ("conv_up2", 64), ("conv_hr", 64), ("conv_last", 1)):
        d[f"{q}.{name}.weight"] = (cout, 64, 3, 3)
        d[f"{q}.{name}.bias"] = (cout,)
    q = "refinenet"   # ResynNet :117-137
    for b, cin in enumerate((6, 12, 12)):
        c = 128
        for i, (a, o) in enumerate(((cin, c // 2), (c // 2, c), (c, 2 * c))):
            d[f"{q}.block{b}.conv0.{i}.0.weight"] = (o, a, 3, 3)
            _bn(d, f"{q}.block{b}.conv0.{i}.1", o)
            d[f"{q}.block{b}.conv0.{i}.2.weight"] = (o,)
        for i in range(6):
            d[f"{q}.block{b}.convblock.{i}.0.weight"] = (2 * c, 2 * c, 3, 3)
            _bn(d, f"{q}.block{b}.convblock.{i}.1", 2 * c)
            d[f"{q}.block{b}.convblock.{i}.2.weight"] = (2 * c,)
        d[f"{q}.block{b}.lastconv.weight"] = (2 * c, 4, 4, 4)
        d[f"{q}.block{b}.lastconv.bias"] = (4,)
    for i, cin in enumerate((3, 32, 32, 32)):   # DegCNN :49-62 (training only)
        _conv(d, f"{q}.degrad.conv{i}", cin, 32)
    d[q + ".degrad.deconv.1.weight"] = (128, 32, 4, 4)
    d[q + ".degrad.deconv.1.bias"] = (32,)
    d[q + ".degrad.deconv.2.weight"] = (32,)
    d[q + ".degrad.deconv.3.weight"] = (3, 32, 3, 3)
    d[q + ".degrad.deconv.3.bias"] = (3,)
    for name in ("context0", "context1"):
        _conv(d, f"{q}.{name}.0", 3, 16)
        _conv(d, f"{q}.{name}.1", 16, 32)
    d[q + ".decode.0.weight"] = (64, 32, 4, 4)
    d[q + ".decode.0.bias"] = (32,)
    d[q + ".decode.1.weight"] = (32, 3, 4, 4)
    d[q + ".decode.1.bias"] = (3,)
    return d
