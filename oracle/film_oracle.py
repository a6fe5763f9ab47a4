"""ORACLE — test infrastructure only.  Never imported by the product path.

CPU restatement (torch-CPU fp32, functional, no nn.Module) of the FILM interpolator the reference's FILM node
executes.  The node loads a TorchScript artifact `film_net_fp32.pt` (vfi_models/film/__init__.py:74) that is NOT
in the tree and cannot be downloaded here; the in-tree vfi_models/film/film_arch.py is the source mirror of the
project that artifact comes from (dajes/frame-interpolation-pytorch v1.0.0, film_arch.py:1-7).  This file
restates film_arch.py and is pinned against THAT (oracle/validate_film_vs_reference.py: bit-exact on seeded
weights).  Against the TorchScript artifact itself parity is UNPINNED.

Restated (file:line in vfi_models/film/film_arch.py):
  conv helper (padding='same', LeakyReLU 0.2)           :784-798
  SubTreeExtractor / FeatureExtractor                    :83-121, :124-162
  FlowEstimator / PyramidFlowEstimator                   :500-543, :546-617
  warp (align_corners=False, border), pyramids, synthesis :655-781
  Fusion                                                 :219-292
  Interpolator.debug_forward (time fixed at 0.5)         :401-455
and the node's per-pair bisection schedule                vfi_models/film/__init__.py:12-42, :63-113
"""
import bisect

import numpy as np
import torch
import torch.nn.functional as F


def conv(sd, key, x, act=True):
    y = F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], padding="same")
    return F.leaky_relu(y, 0.2) if act else y


def build_image_pyramid(image, levels=7):
    pyr = []
    for i in range(levels):
        pyr.append(image)
        if i < levels - 1:
            image = F.avg_pool2d(image, 2, 2)
    return pyr


def subtree(sd, image, n):
    head, pyr = image, []
    for i in range(4):
        p = f"extract.extract_sublevels.convs.{i}."
        head = conv(sd, p + "0.0", head)
        head = conv(sd, p + "1.0", head)
        pyr.append(head)
        if i < n - 1:
            head = F.avg_pool2d(head, kernel_size=2, stride=2)
        if i == n - 1:
            break
    return pyr


def extract(sd, image_pyramid, sub_levels=4):
    n = len(image_pyramid)
    subs = [subtree(sd, image_pyramid[i], min(n - i, sub_levels)) for i in range(n)]
    feats = []
    for i in range(n):
        f = subs[i][0]
        for j in range(1, sub_levels):
            if j <= i:
                f = torch.cat([f, subs[i - j][j]], dim=1)
        feats.append(f)
    return feats


def warp(image, flow):
    flow = -flow.flip(1)
    ls1 = 1 - 1 / flow.shape[3]
    ls2 = 1 - 1 / flow.shape[2]
    nf = flow.permute(0, 2, 3, 1) / torch.tensor([flow.shape[2] * .5, flow.shape[3] * .5], dtype=flow.dtype)[None, None, None]
    g = torch.stack([
        torch.linspace(-ls1, ls1, flow.shape[3], dtype=flow.dtype)[None, None, :] - nf[..., 1],
        torch.linspace(-ls2, ls2, flow.shape[2], dtype=flow.dtype)[None, :, None] - nf[..., 0],
    ], dim=3)
    return F.grid_sample(image, g, mode="bilinear", padding_mode="border", align_corners=False).reshape(image.shape)


def flow_estimator(sd, prefix, a, b):
    net = torch.cat([a, b], dim=1)
    for i in range(3):
        net = conv(sd, f"{prefix}._convs.{i}.0", net)
    net = conv(sd, f"{prefix}._convs.3.0", net)
    return conv(sd, f"{prefix}._convs.4", net, act=False)


def predict_flow(sd, pa, pb):
    levels = len(pa)
    v = flow_estimator(sd, "predict_flow._predictor", pa[-1], pb[-1])
    residuals = [v]
    for i in range(levels - 2, 2, -1):  # len(_predictors) - 1 == 2
        v = F.interpolate(2 * v, size=pa[i].shape[2:4], mode="bilinear")
        vr = flow_estimator(sd, "predict_flow._predictor", pa[i], warp(pb[i], v))
        residuals.insert(0, vr)
        v = vr + v
    for k in range(3):
        i = 2 - k
        v = F.interpolate(2 * v, size=pa[i].shape[2:4], mode="bilinear")
        vr = flow_estimator(sd, f"predict_flow._predictors.{k}", pa[i], warp(pb[i], v))
        residuals.insert(0, vr)
        v = vr + v
    return residuals


def flow_pyramid_synthesis(res):
    flow = res[-1]
    out = [flow]
    for r in res[:-1][::-1]:
        flow = F.interpolate(2 * flow, size=r.shape[2:4], mode="bilinear")
        flow = r + flow
        out.insert(0, flow)
    return out


def fuse(sd, pyramid):
    net = pyramid[-1]
    for k in range(4):
        i = 3 - k
        net = F.interpolate(net, size=pyramid[i].shape[2:4], mode="nearest")
        net = conv(sd, f"fuse.convs.{k}.0", net, act=False)
        net = torch.cat([pyramid[i], net], dim=1)
        net = conv(sd, f"fuse.convs.{k}.1.0", net)
        net = conv(sd, f"fuse.convs.{k}.2.0", net)
    return conv(sd, "fuse.output_conv", net, act=False)


def film_forward(sd, x0, x1, batch_dt=None, return_aux=False):
    """Interpolator.forward: x0,x1 [B,3,H,W] -> [B,3,H,W] (time is hard-wired to 0.5, film_arch.py:427-429)."""
    ip = [build_image_pyramid(x0), build_image_pyramid(x1)]
    fp = [extract(sd, ip[0]), extract(sd, ip[1])]
    fwd_res = predict_flow(sd, fp[0], fp[1])
    bwd_res = predict_flow(sd, fp[1], fp[0])
    fwd_flow = flow_pyramid_synthesis(fwd_res)[:5]
    bwd_flow = flow_pyramid_synthesis(bwd_res)[:5]
    mid = torch.full((x0.shape[0],), .5)
    backward_flow = [f * mid for f in bwd_flow]
    forward_flow = [f * (1 - mid) for f in fwd_flow]
    to_warp = [[torch.cat([a, b], 1) for a, b in zip(ip[k][:5], fp[k][:5])] for k in range(2)]
    fw = [warp(f, fl) for f, fl in zip(to_warp[0], backward_flow)]
    bw = [warp(f, fl) for f, fl in zip(to_warp[1], forward_flow)]
    aligned = [torch.cat([a, b, c, d], 1) for a, b, c, d in zip(fw, bw, backward_flow, forward_flow)]
    out = fuse(sd, aligned)
    if return_aux:
        return out, dict(fwd_flow=fwd_flow, bwd_flow=bwd_flow, feat0=fp[0], aligned=aligned)
    return out


# ---------------------------------------------------------------------------------------------
# node level (vfi_models/film/__init__.py)
# ---------------------------------------------------------------------------------------------

def film_schedule(inter_frames):
    """film/__init__.py:12-42 — order of (left index, right index, new index) midpoint calls for one pair.
    Indices are positions on the ideal grid 0..inter_frames+1."""
    idxes = [0, inter_frames + 1]
    remains = list(range(1, inter_frames + 1))
    splits = torch.linspace(0, 1, inter_frames + 2)
    calls = []
    for _ in range(len(remains)):
        starts = splits[idxes[:-1]]
        ends = splits[idxes[1:]]
        distances = ((splits[None, remains] - starts[:, None]) / (ends[:, None] - starts[:, None]) - .5).abs()
        matrix = torch.argmin(distances).item()
        start_i, step = np.unravel_index(matrix, distances.shape)
        end_i = start_i + 1
        new = remains[step]
        calls.append((idxes[start_i], idxes[end_i], new))
        idxes.insert(bisect.bisect_left(idxes, new), new)
        del remains[step]
    return calls


def film_vfi(sd, frames, multiplier=2, states=None, model=None):
    """Whole-node oracle: frames [N,H,W,C] -> [N_out,H,W,3]; skipped pairs are DROPPED (film/__init__.py:89-90)."""
    x = frames[..., :3].permute(0, 3, 1, 2).float()
    n = len(x)
    if isinstance(multiplier, int):
        ms = [multiplier] * n
    else:
        ms = list(map(int, multiplier))
        ms += [2] * (n - len(ms) - 1)
    model = model or (lambda a, b: film_forward(sd, a, b))
    out = []
    with torch.inference_mode():
        for i in range(n - 1):
            if states is not None and states.is_frame_skipped(i):
                continue
            res = {0: x[i:i + 1], ms[i]: x[i + 1:i + 2]}
            for (l, r, new) in film_schedule(ms[i] - 1):
                res[new] = model(res[l], res[r]).clamp(0, 1).float()
            out.extend(res[k] for k in sorted(res)[:-1])
    out.append(x[-1:])
    return torch.cat(out, 0).permute(0, 2, 3, 1)[..., :3].contiguous()
