"""GMFSS Fortuna (union) checkpoints: key/shape tables of the five networks the node loads
(vfi_models/gmfss_fortuna/__init__.py:11-18, GMFSS_Fortuna_union_arch.py): rife46.pth (IFNet arch "4.6"),
GMFSS_fortuna_flownet.pkl (GMFlow), GMFSS_fortuna_union_metric.pkl (MetricNet), GMFSS_fortuna_union_feat.pkl
(FeatureNet), GMFSS_fortuna_union_fusionnet.pkl (GridNet).  Groundwork for SURVEY.md 8f rank 3 (oracle pinned; the device
path is the next row)."""
from collections import OrderedDict

from .rife_spec import _block_shapes

PARTS = ("ifnet", "flownet", "metricnet", "feat_ext", "fusionnet")
RIFE46_BLOCKS = ((7, 192), (8 + 4, 128), (8 + 4, 96), (8 + 4, 64))    # rife_arch.py:404-408


def ifnet46_shapes():
    d = OrderedDict()
    _block_shapes(d, RIFE46_BLOCKS)
    return d


def gmflow_shapes(c=128, n_layers=6, upsample_factor=4):
    """GMFlow(num_scales=2) (GMFSS_Fortuna_union_arch.py:1157-1199); InstanceNorm2d layers carry no parameters"""
    d = OrderedDict()
    d["backbone.conv1.weight"] = (64, 3, 7, 7)
    cin = 64
    for name, dim, stride in (("layer1", 64, 1), ("layer2", 96, 2), ("layer3", 128, 1)):
        for blk in (0, 1):
            p = f"backbone.{name}.{blk}."
            bi = cin if blk == 0 else dim
            d[p + "conv1.weight"] = (dim, bi, 3, 3)
            d[p + "conv2.weight"] = (dim, dim, 3, 3)
            if blk == 0 and (stride != 1 or bi != dim):
                d[p + "downsample.0.weight"] = (dim, bi, 1, 1)
                d[p + "downsample.0.bias"] = (dim,)
        cin = dim
    d["backbone.conv2.weight"] = (c, 128, 1, 1)
    d["backbone.conv2.bias"] = (c,)
    d["backbone.trident_conv.weight"] = (c, c, 3, 3)
    for i in range(n_layers):
        for part, ffn in (("self_attn", False), ("cross_attn_ffn", True)):
            p = f"transformer.layers.{i}.{part}."
            for n in ("q_proj", "k_proj", "v_proj", "merge"):
                d[p + n + ".weight"] = (c, c)
            d[p + "norm1.weight"] = (c,)
            d[p + "norm1.bias"] = (c,)
            if ffn:
                d[p + "mlp.0.weight"] = (8 * c, 2 * c)
                d[p + "mlp.2.weight"] = (c, 8 * c)
                d[p + "norm2.weight"] = (c,)
                d[p + "norm2.bias"] = (c,)
    for n in ("q_proj", "k_proj"):
        d[f"feature_flow_attn.{n}.weight"] = (c, c)
        d[f"feature_flow_attn.{n}.bias"] = (c,)
    d["upsampler.0.weight"] = (256, 2 + c, 3, 3)
    d["upsampler.0.bias"] = (256,)
    d["upsampler.2.weight"] = (upsample_factor ** 2 * 9, 256, 1, 1)
    d["upsampler.2.bias"] = (upsample_factor ** 2 * 9,)
    return d


def metricnet_shapes():
    d = OrderedDict()
    d["metric_in.weight"] = (64, 14, 3, 3)
    d["metric_in.bias"] = (64,)
    for k in (1, 2, 3):
        d[f"metric_net{k}.0.weight"] = (1,)
        d[f"metric_net{k}.1.weight"] = (64, 64, 3, 3)
        d[f"metric_net{k}.1.bias"] = (64,)
    d["metric_out.0.weight"] = (1,)
    d["metric_out.1.weight"] = (2, 64, 3, 3)
    d["metric_out.1.bias"] = (2,)
    return d


def _pair(d, p, cin, cout, transposed=False):
    """Sequential(PReLU, Conv2d | ConvTranspose2d(4,2,1), PReLU, Conv2d)"""
    d[p + "0.weight"] = (1,)
    d[p + "1.weight"] = (cin, cout, 4, 4) if transposed else (cout, cin, 3, 3)
    d[p + "1.bias"] = (cout,)
    d[p + "2.weight"] = (1,)
    d[p + "3.weight"] = (cout, cout, 3, 3)
    d[p + "3.bias"] = (cout,)


def featurenet_shapes():
    d = OrderedDict()
    for k, (cin, cout) in enumerate(((3, 64), (64, 128), (128, 192)), 1):
        _pair(d, f"block{k}.", cin, cout)
    return d


def gridnet_shapes(cin=9, c1=128, c2=256, c3=384, cout=3, head="head0"):
    """GridNet: head inputs 9 (union: I1t, rife, I2t; key residual_model_head0) or 12 (base model: img0, I1t, I2t, img1; key
    residual_model_head) / 128 / 256 / 384 channels (GMFSS_Fortuna_union_arch.py:1582-1637, GMFSS_Fortuna_arch.py:1583-1638)"""
    d = OrderedDict()
    for name, (a, b) in ((head, (cin, 64)), ("head1", (c1, 64)), ("head2", (c2, 128)), ("head3", (c3, 192)),
                         ("01", (64, 64)), ("04", (64, 64)), ("05", (64, 64))):
        _pair(d, f"residual_model_{name}.", a, b)
    p = "residual_model_tail."
    d[p + "conv_before_upsample.0.weight"] = (64, 64, 3, 3)
    d[p + "conv_before_upsample.0.bias"] = (64,)
    d[p + "conv_before_upsample.1.weight"] = (1,)
    d[p + "upsample.0.weight"] = (256, 64, 3, 3)
    d[p + "upsample.0.bias"] = (256,)
    d[p + "conv_last.weight"] = (cout, 64, 3, 3)
    d[p + "conv_last.bias"] = (cout,)
    for name, c in (("11", 128), ("14", 128), ("15", 128), ("21", 192), ("24", 192), ("25", 192)):
        _pair(d, f"residual_model_{name}.", c, c)
    for name, (a, b) in (("10", (64, 128)), ("20", (128, 192)), ("11", (64, 128)), ("21", (128, 192))):
        _pair(d, f"downsample_model_{name}.", a, b)
    for name, (a, b) in (("04", (128, 64)), ("14", (192, 128)), ("05", (128, 64)), ("15", (192, 128))):
        _pair(d, f"upsample_model_{name}.", a, b, transposed=True)
    return d


def gmfss_union_shapes():
    return {"ifnet": ifnet46_shapes(), "flownet": gmflow_shapes(), "metricnet": metricnet_shapes(), "feat_ext": featurenet_shapes(),
            "fusionnet": gridnet_shapes()}


def gmfss_base_shapes():
    """"GMFSS_fortuna" (gmfss_fortuna/__init__.py:19-24, GMFSS_Fortuna_arch.py): no IFNet, 12-channel GridNet head"""
    return {"flownet": gmflow_shapes(), "metricnet": metricnet_shapes(), "feat_ext": featurenet_shapes(),
            "fusionnet": gridnet_shapes(cin=12, head="head")}


def gmfss_shapes(variant):
    return {"union": gmfss_union_shapes, "base": gmfss_base_shapes}[variant]()
