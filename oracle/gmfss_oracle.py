"""ORACLE — test infrastructure only; the product path never imports this module.

CPU restatement (torch, fp32) of the reference's GMFSS Fortuna path, written as plain functions
over the five checkpoints' state_dicts:
    vfi_models/gmfss_fortuna/__init__.py:27-78,110-143          (CommonModelInference.forward, the node)
    vfi_models/gmfss_fortuna/GMFSS_Fortuna_union_arch.py        (Model.reuse / Model.inference :1726-1857, GMFlow :1157-1372
        with its backbone :68-312, Swin-style transformer :315-686, matching :806-913, flow propagation :688-803, convex
        up-sampling :1220-1260; MetricNet :1375-1467; FeatureNet :1470-1500; GridNet :1503-1688)
    vfi_models/rife/rife_arch.py:465-732                        (IFNet arch "4.6": 4.7's IFBlocks without the encoder, flow
        AND mask accumulated, inputs (img0, img1, timestep[, mask]))
    vfi_models/ops/cupy_ops/softsplat.py:382-435                (softsplat(..., "soft"): exp-metric weighting around the
        summation splat; the splat itself is the C restatement oracle/m2m_ops.c — the reference has no CPU path for it)
Status: pinned bit-exactly against the reference modules on seeded weights by oracle/validate_gmfss_vs_reference.py
(oracle/VALIDATION_GMFSS.log; the summation splat via the same C restatement on both sides, as for M2M).
``sds`` below is a dict {"ifnet", "flownet", "metricnet", "feat_ext", "fusionnet"} of state_dicts
(gmfss_fortuna/__init__.py:11-18).
"""
import math

import torch
import torch.nn.functional as F

from . import rife_oracle


# ---- GMFlow: backbone -----------------------------------------------------------------------------------------------
def _residual(sd, p, x, stride):
    """ResidualBlock_class.forward (:207-215): conv-IN-relu x2 (+ 1x1 conv-IN shortcut when the shape changes)"""
    y = F.relu(F.instance_norm(F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)))
    y = F.relu(F.instance_norm(F.conv2d(y, sd[p + "conv2.weight"], None, 1, 1)))
    if p + "downsample.0.weight" in sd:
        x = F.instance_norm(F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride))
    return F.relu(x + y)


def backbone(sd, x, p="backbone."):
    """CNNEncoder.forward (:296-312) with num_output_scales=2: features at 1/4 and 1/8 (high to low resolution)"""
    x = F.relu(F.instance_norm(F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 1)):
        x = _residual(sd, f"{p}{name}.0.", x, stride)
        x = _residual(sd, f"{p}{name}.1.", x, 1)
    x = F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    w = sd[p + "trident_conv.weight"]          # one weight, two strides (MultiScaleTridentConv :122-162, bias=False)
    return [F.conv2d(x, w, None, 1, 1), F.conv2d(x, w, None, 2, 1)]


# ---- GMFlow: window helpers (:1059-1120) ----------------------------------------------------------------------------
def _to_windows(t, k, channel_last):
    if channel_last:
        b, h, w, c = t.shape
        return t.view(b, k, h // k, k, w // k, c).permute(0, 1, 3, 2, 4, 5).reshape(b * k * k, h // k, w // k, c)
    b, c, h, w = t.shape
    return t.view(b, c, k, h // k, k, w // k).permute(0, 2, 4, 1, 3, 5).reshape(b * k * k, c, h // k, w // k)


def _from_windows(t, k, channel_last):
    if channel_last:
        b, h, w, c = t.shape
        nb = b // k // k
        return t.view(nb, k, k, h, w, c).permute(0, 1, 3, 2, 4, 5).contiguous().view(nb, k * h, k * w, c)
    b, c, h, w = t.shape
    nb = b // k // k
    return t.view(nb, k, k, c, h, w).permute(0, 3, 1, 4, 2, 5).contiguous().view(nb, c, k * h, k * w)


def _sine_position(x, temperature=10000.0):
    """PositionEmbeddingSine.forward (:1032-1056), num_pos_feats = C/2, normalize=True, scale 2*pi"""
    b, c, h, w = x.shape
    npf = c // 2
    ones = torch.ones((b, h, w))
    y_embed = ones.cumsum(1, dtype=torch.float32)
    x_embed = ones.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + 1e-6) * (2 * math.pi)
    x_embed = x_embed / (x_embed[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / npf)
    px = x_embed[:, :, :, None] / dim_t
    py = y_embed[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def _add_position(f0, f1, splits):
    """feature_add_position (:1134-1154): the embedding restarts in every attention window"""
    if splits > 1:
        a, b = _to_windows(f0, splits, False), _to_windows(f1, splits, False)
        pos = _sine_position(a)
        return _from_windows(a + pos, splits, False), _from_windows(b + pos, splits, False)
    pos = _sine_position(f0)
    return f0 + pos, f1 + pos


def _shift_mask(h, w, wh, ww):
    """generate_shift_window_attn_mask (:326-364): tokens of a shifted window that come from different sides of the
    roll seam must not attend to each other (-100 before the softmax)"""
    def region(n, win):
        r = torch.zeros(n)
        r[n - win:n - win // 2] = 1
        r[n - win // 2:] = 2
        return r

    label = (region(h, wh)[:, None] * 3 + region(w, ww)[None, :]).view(1, h, w, 1)
    win = _to_windows(label, w // ww, True).view(-1, wh * ww)
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return diff.masked_fill(diff != 0, -100.0).masked_fill(diff == 0, 0.0)


def _window_attention(q, k, v, splits, shifted, h, w, mask):
    """single_head_split_window_attention (:367-436)"""
    b, _, c = q.shape
    wh, ww = h // splits, w // splits
    q, k, v = (t.view(b, h, w, c) for t in (q, k, v))
    if shifted:
        q, k, v = (torch.roll(t, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2)) for t in (q, k, v))
    q, k, v = (_to_windows(t, splits, True) for t in (q, k, v))
    nb = b * splits * splits
    scores = torch.matmul(q.view(nb, -1, c), k.view(nb, -1, c).permute(0, 2, 1)) / (c ** 0.5)
    if shifted:
        scores += mask.repeat(b, 1, 1)
    out = torch.matmul(torch.softmax(scores, dim=-1), v.view(nb, -1, c))
    out = _from_windows(out.view(nb, wh, ww, c), splits, True)
    if shifted:
        out = torch.roll(out, shifts=(wh // 2, ww // 2), dims=(1, 2))
    return out.view(b, -1, c)


def _attention_layer(sd, p, source, target, h, w, mask, splits, shifted, ffn):
    """TransformerLayer.forward (:479-523)"""
    c = source.shape[-1]
    q = F.linear(source, sd[p + "q_proj.weight"])
    k = F.linear(target, sd[p + "k_proj.weight"])
    v = F.linear(target, sd[p + "v_proj.weight"])
    if splits > 1:
        msg = _window_attention(q, k, v, splits, shifted, h, w, mask)
    else:
        msg = torch.matmul(torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / (c ** 0.5), dim=2), v)
    msg = F.layer_norm(F.linear(msg, sd[p + "merge.weight"]), (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    if ffn:
        msg = F.linear(F.gelu(F.linear(torch.cat([source, msg], dim=-1), sd[p + "mlp.0.weight"])), sd[p + "mlp.2.weight"])
        msg = F.layer_norm(msg, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    return source + msg


def transformer(sd, f0, f1, splits, p="transformer.", n_layers=6):
    """FeatureTransformer.forward (:628-685): both directions as one batch, self- then cross-attention (+FFN) per block"""
    b, c, h, w = f0.shape
    f0 = f0.flatten(-2).permute(0, 2, 1)
    f1 = f1.flatten(-2).permute(0, 2, 1)
    mask = _shift_mask(h, w, h // splits, w // splits) if splits > 1 else None
    a = torch.cat((f0, f1), dim=0)
    o = torch.cat((f1, f0), dim=0)
    for i in range(n_layers):
        shifted = i % 2 == 1
        q = f"{p}layers.{i}."
        a = _attention_layer(sd, q + "self_attn.", a, a, h, w, mask, splits, shifted, False)
        a = _attention_layer(sd, q + "cross_attn_ffn.", a, o, h, w, mask, splits, shifted, True)
        o = torch.cat(a.chunk(chunks=2, dim=0)[::-1], dim=0)
    f0, f1 = a.chunk(chunks=2, dim=0)
    return f0.view(b, h, w, c).permute(0, 3, 1, 2).contiguous(), f1.view(b, h, w, c).permute(0, 3, 1, 2).contiguous()


# ---- GMFlow: matching, propagation, up-sampling ---------------------------------------------------------------------
def _pixel_grid(b, h, w):
    """coords_grid (:916-932): [B,2,H,W], channel 0 = x"""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([x, y], dim=0).float()[None].repeat(b, 1, 1, 1)


def global_match(f0, f1):
    """global_correlation_softmax (:806-843), forward flow only: softmax over all target pixels, expected coordinate"""
    b, c, h, w = f0.shape
    corr = torch.matmul(f0.view(b, c, -1).permute(0, 2, 1), f1.view(b, c, -1)).view(b, h, w, h, w) / (c ** 0.5)
    init = _pixel_grid(b, h, w)
    grid = init.view(b, 2, -1).permute(0, 2, 1)
    prob = F.softmax(corr.view(b, h * w, h * w), dim=-1)
    return torch.matmul(prob, grid).view(b, h, w, 2).permute(0, 3, 1, 2) - init


def local_match(f0, f1, radius):
    """local_correlation_softmax (:846-913): (2r+1)^2 window around each pixel, out-of-image taps masked with -1e9"""
    b, c, h, w = f0.shape
    init = _pixel_grid(b, h, w)
    coords = init.view(b, 2, -1).permute(0, 2, 1)
    n = 2 * radius + 1
    gx, gy = torch.meshgrid([torch.linspace(-radius, radius, n), torch.linspace(-radius, radius, n)], indexing="ij")
    window = torch.stack((gx, gy), -1).transpose(0, 1).float().reshape(-1, 2).repeat(b, 1, 1, 1)
    sample = coords.unsqueeze(-2) + window
    valid = (sample[..., 0] >= 0) & (sample[..., 0] < w) & (sample[..., 1] >= 0) & (sample[..., 1] < h)
    half = torch.Tensor([(w - 1) / 2.0, (h - 1) / 2.0]).float()
    feat = F.grid_sample(f1, (sample - half) / half, padding_mode="zeros", align_corners=True).permute(0, 2, 1, 3)
    corr = torch.matmul(f0.permute(0, 2, 3, 1).view(b, h * w, 1, c), feat).view(b, h * w, -1) / (c ** 0.5)
    corr[~valid] = -1e9
    prob = F.softmax(corr, -1)
    return torch.matmul(prob.unsqueeze(-2), sample).squeeze(-2).view(b, h, w, 2).permute(0, 3, 1, 2) - init


def _flow_sample(feature, flow):
    """flow_warp -> bilinear_sample (:955-991): zeros padding, align_corners=True, coordinates pixel + flow"""
    b, c, h, w = feature.shape
    g = _pixel_grid(b, h, w) + flow
    grid = torch.stack([2 * g[:, 0] / (w - 1) - 1, 2 * g[:, 1] / (h - 1) - 1], dim=-1)
    return F.grid_sample(feature, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def propagate(sd, f0, flow, radius, p="feature_flow_attn."):
    """FeatureFlowAttention.forward (:708-803): flow re-estimated as a feature-similarity weighted mean of the flow field
    (global when radius <= 0, else over a (2r+1)^2 window); the key is projected from the PROJECTED query (:728-735)"""
    b, c, h, w = f0.shape
    qw, qb, kw, kb = (sd[p + n] for n in ("q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias"))
    tokens = f0.view(b, c, h * w).permute(0, 2, 1)
    if radius <= 0:
        q = F.linear(tokens, qw, qb)
        k = F.linear(q, kw, kb)
        v = flow.view(b, flow.size(1), h * w).permute(0, 2, 1)
        prob = torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / (c ** 0.5), dim=-1)
        return torch.matmul(prob, v).view(b, h, w, v.size(-1)).permute(0, 3, 1, 2)
    n = 2 * radius + 1
    q = F.linear(tokens, qw, qb).reshape(b * h * w, 1, c)
    k = F.linear(tokens, kw, kb).permute(0, 2, 1).reshape(b, c, h, w)
    kwin = F.unfold(k, kernel_size=n, padding=radius).view(b, c, n * n, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, c, n * n)
    vwin = F.unfold(flow, kernel_size=n, padding=radius).view(b, 2, n * n, h, w).permute(0, 3, 4, 2, 1).reshape(b * h * w, n * n, 2)
    prob = torch.softmax(torch.matmul(q, kwin) / (c ** 0.5), dim=-1)
    return torch.matmul(prob, vwin).view(b, h, w, 2).permute(0, 3, 1, 2).contiguous()


def convex_upsample(sd, flow, feature, factor=4, p="upsampler."):
    """GMFlow.upsample_flow, convex branch (:1237-1258): each fine pixel = softmax-weighted mean of the 3x3 coarse flows"""
    x = F.relu(F.conv2d(torch.cat((flow, feature), dim=1), sd[p + "0.weight"], sd[p + "0.bias"], 1, 1))
    mask = F.conv2d(x, sd[p + "2.weight"], sd[p + "2.bias"])
    b, fc, h, w = flow.shape
    mask = torch.softmax(mask.view(b, 1, 9, factor, factor, h, w), dim=2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(b, fc, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(b, fc, factor * h, factor * w)


def gmflow(sd, img0, img1):
    """GMFlow.forward (:1262-1372) as Model.reuse calls it: 2 scales, attn splits [2, 8], global matching then a radius-4
    local refinement, global then radius-1 propagation, forward flow only; returns the flow at the input resolution."""
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img0, img1 = (img0 - mean) / std, (img1 - mean) / std
    feats = backbone(sd, torch.cat((img0, img1), dim=0))[::-1]            # low to high resolution
    flow = None
    for scale, (splits, corr_radius, prop_radius) in enumerate(((2, -1, -1), (8, 4, 1))):
        f0, f1 = torch.chunk(feats[scale], 2, 0)
        if scale > 0:
            flow = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2
            f1 = _flow_sample(f1, flow)
        f0, f1 = _add_position(f0, f1, splits)
        f0, f1 = transformer(sd, f0, f1, splits)
        pred = global_match(f0, f1) if corr_radius == -1 else local_match(f0, f1, corr_radius)
        flow = flow + pred if flow is not None else pred
        flow = propagate(sd, f0, flow, prop_radius)
    return convex_upsample(sd, flow, f0)


# ---- MetricNet / FeatureNet / GridNet -------------------------------------------------------------------------------
def _backwarp_zeros(x, flow):
    """backwarp (:1375-1417): grid_sample(zeros, align_corners=True) at linspace grid + flow/((size-1)/2)"""
    b, _, h, w = flow.shape
    hor = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, -1).repeat(1, 1, h, 1)
    ver = torch.linspace(-1.0, 1.0, h).view(1, 1, -1, 1).repeat(1, 1, 1, w)
    f = torch.cat([flow[:, 0:1] / ((x.shape[3] - 1.0) / 2.0), flow[:, 1:2] / ((x.shape[2] - 1.0) / 2.0)], 1)
    return F.grid_sample(x, (torch.cat([hor, ver], 1) + f).permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros", align_corners=True)


def _occlusion(fwd, bwd, alpha=0.01, beta=0.5):
    """forward_backward_consistency_check (:994-1012)"""
    mag = torch.norm(fwd, dim=1) + torch.norm(bwd, dim=1)
    d_fwd = torch.norm(fwd + _flow_sample(bwd, fwd), dim=1)
    d_bwd = torch.norm(bwd + _flow_sample(fwd, bwd), dim=1)
    thr = alpha * mag + beta
    return (d_fwd > thr).float(), (d_bwd > thr).float()


def metricnet(sd, img0, img1, flow01, flow10):
    """MetricNet.forward (:1429-1467) -> (metric0, metric1) in [-10, 10]"""
    m0 = F.l1_loss(img0, _backwarp_zeros(img1, flow01), reduction="none").mean([1], True)
    m1 = F.l1_loss(img1, _backwarp_zeros(img0, flow10), reduction="none").mean([1], True)
    occ_f, occ_b = _occlusion(flow01, flow10)
    h, w = flow01.shape[2:]
    n01 = torch.cat([flow01[:, 0:1] / ((w - 1.0) / 2.0), flow01[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    n10 = torch.cat([flow10[:, 0:1] / ((w - 1.0) / 2.0), flow10[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    x = torch.cat((img0, img1, -m0, -m1, n01, n10, occ_f.unsqueeze(1), occ_b.unsqueeze(1)), 1)
    feat = F.conv2d(x, sd["metric_in.weight"], sd["metric_in.bias"], 1, 1)
    for k in (1, 2, 3):
        feat = F.conv2d(F.prelu(feat, sd[f"metric_net{k}.0.weight"]), sd[f"metric_net{k}.1.weight"], sd[f"metric_net{k}.1.bias"], 1, 1) + feat
    out = F.conv2d(F.prelu(feat, sd["metric_out.0.weight"]), sd["metric_out.1.weight"], sd["metric_out.1.bias"], 1, 1)
    out = torch.tanh(out) * 10
    return out[:, :1], out[:, 1:2]


def _prelu_conv_pair(sd, p, x, stride1, stride2=1, transposed=False):
    """Sequential(PReLU, Conv2d|ConvTranspose2d(4,2,1), PReLU, Conv2d): ResidualBlock / DownsampleBlock / UpsampleBlock /
    FeatureNet.block (:1470-1561)"""
    x = F.prelu(x, sd[p + "0.weight"])
    if transposed:
        x = F.conv_transpose2d(x, sd[p + "1.weight"], sd[p + "1.bias"], 2, 1)
    else:
        x = F.conv2d(x, sd[p + "1.weight"], sd[p + "1.bias"], stride1, 1)
    return F.conv2d(F.prelu(x, sd[p + "2.weight"]), sd[p + "3.weight"], sd[p + "3.bias"], stride2, 1)


def featurenet(sd, x):
    """FeatureNet.forward (:1494-1500): 3 levels at 1/2, 1/4, 1/8 with 64, 128, 192 channels"""
    x1 = _prelu_conv_pair(sd, "block1.", x, 2)
    x2 = _prelu_conv_pair(sd, "block2.", x1, 2)
    x3 = _prelu_conv_pair(sd, "block3.", x2, 2)
    return x1, x2, x3


def gridnet(sd, x, x1, x2, x3):
    """GridNet.forward (:1639-1688): 3-row grid of residual / down / up blocks, PixelShuffle tail"""
    def res(name, t):
        return _prelu_conv_pair(sd, f"residual_model_{name}.", t, 1)

    def down(name, t):
        return _prelu_conv_pair(sd, f"downsample_model_{name}.", t, 2)

    def up(name, t):
        return _prelu_conv_pair(sd, f"upsample_model_{name}.", t, 2, transposed=True)

    x00 = res("head0" if "residual_model_head0.0.weight" in sd else "head", x) + res("head1", x1)
    x01 = res("01", x00) + x00
    x10 = down("10", x00) + res("head2", x2)
    x20 = down("20", x10) + res("head3", x3)
    x11 = (res("11", x10) + x10) + down("11", x01)
    x21 = (res("21", x20) + x20) + down("21", x11)
    x24 = res("24", x21) + x21
    x25 = res("25", x24) + x24
    x14 = up("14", x24) + (res("14", x11) + x11)
    x04 = up("04", x14) + (res("04", x01) + x01)
    x15 = up("15", x25) + (res("15", x14) + x14)
    x05 = up("05", x15) + (res("05", x04) + x04)
    p = "residual_model_tail."
    t = F.prelu(F.conv2d(x05, sd[p + "conv_before_upsample.0.weight"], sd[p + "conv_before_upsample.0.bias"], 1, 1),
                sd[p + "conv_before_upsample.1.weight"])
    t = F.pixel_shuffle(F.conv2d(t, sd[p + "upsample.0.weight"], sd[p + "upsample.0.bias"], 1, 1), 2)
    return F.conv2d(t, sd[p + "conv_last.weight"], sd[p + "conv_last.bias"], 1, 1)


# ---- IFNet arch 4.6 and the soft splat ------------------------------------------------------------------------------
def ifnet46_forward(sd, img0, img1, timestep, scale_list=(8, 4, 2, 1)):
    """IFNet("4.6").forward (rife_arch.py:465-732) as GMFSS calls it: float timestep, ensemble off, fastmode on.
    Same IFBlocks as 4.7 (rife_oracle.ifblock); no encoder, flow and mask are accumulated over the blocks."""
    img0, img1 = torch.clamp(img0, 0, 1), torch.clamp(img1, 0, 1)
    n, c, h, w = img0.shape
    ph, pw = ((h - 1) // 64 + 1) * 64, ((w - 1) // 64 + 1) * 64
    img0, img1 = F.pad(img0, (0, pw - w, 0, ph - h)), F.pad(img1, (0, pw - w, 0, ph - h))
    x = torch.cat((img0, img1), 1)
    t = (x[:, :1].clone() * 0 + 1) * timestep
    flow = mask = None
    w0, w1 = img0, img1
    for i in range(4):
        if flow is None:
            flow, mask = rife_oracle.ifblock(sd, f"block{i}.", torch.cat((img0[:, :3], img1[:, :3], t), 1), None, scale_list[i])
        else:
            fd, md = rife_oracle.ifblock(sd, f"block{i}.", torch.cat((w0[:, :3], w1[:, :3], t, mask), 1), flow, scale_list[i])
            flow = flow + fd
            mask = mask + md
        w0, w1 = rife_oracle.warp(img0, flow[:, :2]), rife_oracle.warp(img1, flow[:, 2:4])
    m = torch.sigmoid(mask)
    return (w0 * m + w1 * (1 - m))[:, :, :h, :w]


def splat_sum(x, flow):
    """softsplat_func.forward: the C restatement of the CUDA kernel text (oracle/m2m_ops.c)"""
    from . import m2m_oracle

    return torch.from_numpy(m2m_oracle.softsplat_sum(x.detach().contiguous().numpy(), flow.detach().contiguous().numpy()))


def softsplat_soft(x, flow, metric):
    """softsplat(tenIn, tenFlow, tenMetric, "soft") (cupy_ops/softsplat.py:382-435)"""
    out = splat_sum(torch.cat([x * metric.exp(), metric.exp()], 1), flow)
    return out[:, :-1] / (out[:, -1:] + 0.0000001)


# ---- the model ------------------------------------------------------------------------------------------------------
def reuse(sds, img0, img1):
    """Model.reuse (:1726-1782) with scale == 1: features of both frames, both flows at half resolution, splat metrics"""
    feats0 = featurenet(sds["feat_ext"], img0)
    feats1 = featurenet(sds["feat_ext"], img1)
    h0 = F.interpolate(img0, scale_factor=0.5, mode="bilinear", align_corners=False)
    h1 = F.interpolate(img1, scale_factor=0.5, mode="bilinear", align_corners=False)
    flow01 = gmflow(sds["flownet"], h0, h1)
    flow10 = gmflow(sds["flownet"], h1, h0)
    metric0, metric1 = metricnet(sds["metricnet"], h0, h1, flow01, flow10)
    return flow01, flow10, metric0, metric1, feats0, feats1


def inference(sds, img0, img1, state, timestep):
    """Model.inference (:1784-1857): 8 soft splats (half-res images + 3 feature levels, both directions), RIFE 4.6 on
    the half-res pair, GridNet fusion; clamp(0, 1)"""
    flow01, flow10, metric0, metric1, feats0, feats1 = state
    f1t, f2t = timestep * flow01, (1 - timestep) * flow10
    z1t, z2t = timestep * metric0, (1 - timestep) * metric1
    h0 = F.interpolate(img0, scale_factor=0.5, mode="bilinear", align_corners=False)
    h1 = F.interpolate(img1, scale_factor=0.5, mode="bilinear", align_corners=False)
    i1t = softsplat_soft(h0, f1t, z1t)
    i2t = softsplat_soft(h1, f2t, z2t)
    # union model: IFNet 4.6 between the two splats; base model (GMFSS_Fortuna_arch.py:1843-1849): the two half-res images
    head = [i1t, ifnet46_forward(sds["ifnet"], h0, h1, timestep), i2t] if "ifnet" in sds else [h0, i1t, i2t, h1]
    levels = []
    for lvl, s in enumerate((1.0, 0.5, 0.25)):
        if s == 1.0:
            fa, za, fb, zb = f1t, z1t, f2t, z2t
        else:
            fa = F.interpolate(f1t, scale_factor=s, mode="bilinear", align_corners=False) * s
            za = F.interpolate(z1t, scale_factor=s, mode="bilinear", align_corners=False)
            fb = F.interpolate(f2t, scale_factor=s, mode="bilinear", align_corners=False) * s
            zb = F.interpolate(z2t, scale_factor=s, mode="bilinear", align_corners=False)
        levels.append(torch.cat([softsplat_soft(feats0[lvl], fa, za), softsplat_soft(feats1[lvl], fb, zb)], dim=1))
    out = gridnet(sds["fusionnet"], torch.cat(head, dim=1), *levels)
    return torch.clamp(out, 0, 1)


def gmfss_forward(sds, i0, i1, timestep):
    """CommonModelInference.forward (gmfss_fortuna/__init__.py:41-78) with scale == 1: zero-pad to x64, reuse, inference, crop"""
    n, c, h, w = i0.shape
    ph, pw = ((h - 1) // 64 + 1) * 64, ((w - 1) // 64 + 1) * 64
    i0, i1 = F.pad(i0, (0, pw - w, 0, ph - h)), F.pad(i1, (0, pw - w, 0, ph - h))
    return inference(sds, i0, i1, reuse(sds, i0, i1), timestep)[:, :, :h, :w]


def gmfss_vfi(sds, frames, multiplier=2, states=None):
    """Node-level oracle (gmfss_fortuna/__init__.py:110-143 + generic_frame_loop, vfi_utils.py:149-389, int multiplier);
    frames [N,H,W,C] fp32 -> [N',H,W,3]"""
    x = frames[..., :3].permute(0, 3, 1, 2).float()
    out = []
    with torch.inference_mode():
        for i in range(len(x) - 1):
            out.append(x[i:i + 1])
            if states is not None and states.is_frame_skipped(i):
                continue
            for k in range(1, multiplier):
                out.append(gmfss_forward(sds, x[i:i + 1], x[i + 1:i + 2], k / multiplier))
        out.append(x[-1:])
    return torch.cat(out, 0).permute(0, 2, 3, 1).contiguous()
