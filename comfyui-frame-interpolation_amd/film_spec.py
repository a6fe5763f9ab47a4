"""Checkpoint layout of the FILM interpolator (film_net_fp32.pt, TorchScript of dajes/frame-interpolation-pytorch).

Key names and shapes are those of ``Interpolator`` in the reference's source mirror
(vfi_models/film/film_arch.py:366-399 and the sub-modules it builds); a TorchScript archive exposes the same
``state_dict()``.  Conv2d weights are [Cout, Cin, kh, kw]."""
from collections import OrderedDict

FILTERS = 64
SUB_LEVELS = 4
PYRAMID_LEVELS = 7
FUSION_LEVELS = 5
FLOW_FILTERS = (32, 64, 128, 256)


def feat_channels(level):
    """channels of the cascaded feature pyramid at `level`: 64, 192, 448, 960, 960, ..."""
    return sum(FILTERS << j for j in range(min(level, SUB_LEVELS - 1) + 1))


def film_shapes():
    d = OrderedDict()
    cin = 3
    for i in range(SUB_LEVELS):
        c = FILTERS << i
        p = f"extract.extract_sublevels.convs.{i}."
        d[p + "0.0.weight"] = (c, cin, 3, 3)
        d[p + "0.0.bias"] = (c,)
        d[p + "1.0.weight"] = (c, c, 3, 3)
        d[p + "1.0.bias"] = (c,)
        cin = c

    def estimator(prefix, in_ch, nf):
        c = in_ch
        for i in range(3):
            d[f"{prefix}._convs.{i}.0.weight"] = (nf, c, 3, 3)
            d[f"{prefix}._convs.{i}.0.bias"] = (nf,)
            c = nf
        d[f"{prefix}._convs.3.0.weight"] = (nf // 2, nf, 1, 1)
        d[f"{prefix}._convs.3.0.bias"] = (nf // 2,)
        d[f"{prefix}._convs.4.weight"] = (2, nf // 2, 1, 1)
        d[f"{prefix}._convs.4.bias"] = (2,)

    # in_channels per level 0..3: 128, 384, 896, 1920 ; state_dict order: shared coarse predictor first,
    # then the specialised ones coarse-to-fine (film_arch.py:562-563)
    ins = [2 * feat_channels(l) for l in range(4)]
    estimator("predict_flow._predictor", ins[3], FLOW_FILTERS[3])
    for k, lvl in enumerate((2, 1, 0)):
        estimator(f"predict_flow._predictors.{k}", ins[lvl], FLOW_FILTERS[lvl])
    d["fuse.output_conv.weight"] = (3, FILTERS, 1, 1)
    d["fuse.output_conv.bias"] = (3,)
    for k in range(4):
        lvl = 3 - k
        nf = FILTERS << min(lvl, 3)
        below = 2 * (3 + feat_channels(lvl + 1)) + 4 if k == 0 else FILTERS << min(lvl + 1, 3)
        skip = 2 * (3 + feat_channels(lvl)) + 4
        d[f"fuse.convs.{k}.0.weight"] = (nf, below, 2, 2)
        d[f"fuse.convs.{k}.0.bias"] = (nf,)
        d[f"fuse.convs.{k}.1.0.weight"] = (nf, skip + nf, 3, 3)
        d[f"fuse.convs.{k}.1.0.bias"] = (nf,)
        d[f"fuse.convs.{k}.2.0.weight"] = (nf, nf, 3, 3)
        d[f"fuse.convs.{k}.2.0.bias"] = (nf,)
    return d


def check_state_dict(sd):
    want = film_shapes()
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    if missing or unexpected:
        raise RuntimeError(f"Error(s) in loading state_dict for FILM Interpolator: Missing key(s): {missing}. "
                           f"Unexpected key(s): {unexpected}.")
    for k, shp in want.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shp)}")
