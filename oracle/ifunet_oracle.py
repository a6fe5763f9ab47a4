"""ORACLE — test infrastructure only; the product path never imports this module.

CPU restatement (torch, fp32) of the reference's IFUNet path (second half):
    vfi_models/ifunet/__init__.py:32-59       (the node: model(frame_0, frame_1, timestep=, scale=, ensemble=) per task)
    vfi_models/ifunet/IFUNet_arch.py          (IFUNetModel.forward :753-766; IFUNet :654-743 with its CBAM U-Net FeatureNet
        :521-597 and the convex-up-sampling IFBlocks :600-651; RRDBNet mask fusion :209-328; ResynNet refinement :75-193;
        warp :331-361; CBAM :364-503)
written as plain functions over the checkpoint's state_dict (eval mode: BatchNorm uses its running statistics, Dropout2d is
the identity).  Pinned bit-exactly against the reference module and node on seeded weights by
oracle/validate_ifunet_vs_reference.py (oracle/VALIDATION_IFUNET.log).
"""
import torch
import torch.nn.functional as F

from .rife_oracle import warp   # same function as IFUNet_arch.warp (:331-361): border, align_corners=True


def _cp(sd, p, x, stride=1, k=3):
    """conv(): Conv2d(bias) + PReLU(c) (:18-30,506-518)"""
    return F.prelu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride, k // 2), sd[p + ".1.weight"])


def _cbp(sd, p, x, stride=1):
    """conv_bn(): Conv2d(no bias) + BatchNorm2d (eval) + PReLU(c) (:33-46)"""
    y = F.conv2d(x, sd[p + ".0.weight"], None, stride, 1)
    y = F.batch_norm(y, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"], False, 0.1, 1e-5)
    return F.prelu(y, sd[p + ".2.weight"])


def cbam(sd, p, x):
    """CBAM.forward (:485-503): channel gate (avg + max pooled MLP) then spatial gate (7x7 conv + BN on [max_c, mean_c])"""
    h, w = x.shape[2:]
    att = None
    for pooled in (F.avg_pool2d(x, (h, w), stride=(h, w)), F.max_pool2d(x, (h, w), stride=(h, w))):
        v = pooled.view(pooled.size(0), -1)
        v = F.linear(F.relu(F.linear(v, sd[p + ".ChannelGate.mlp.1.weight"], sd[p + ".ChannelGate.mlp.1.bias"])),
                     sd[p + ".ChannelGate.mlp.3.weight"], sd[p + ".ChannelGate.mlp.3.bias"])
        att = v if att is None else att + v
    x = x * torch.sigmoid(att).unsqueeze(2).unsqueeze(3).expand_as(x)
    comp = torch.cat((torch.max(x, 1)[0].unsqueeze(1), torch.mean(x, 1).unsqueeze(1)), dim=1)
    q = p + ".SpatialGate.spatial."
    s = F.conv2d(comp, sd[q + "conv.weight"], None, 1, 3)
    s = F.batch_norm(s, sd[q + "bn.running_mean"], sd[q + "bn.running_var"], sd[q + "bn.weight"], sd[q + "bn.bias"], False, 0.01, 1e-5)
    return x * torch.sigmoid(s)


def _unet_conv(sd, p, x):
    """UNetConv.forward (:532-537)"""
    x = _cp(sd, p + ".conv2", _cp(sd, p + ".conv1", x, 2))
    return cbam(sd, p + ".cbam", x) if p + ".cbam.ChannelGate.mlp.1.weight" in sd else x


def _up_conv(sd, p, x1, x2):
    """UpConv.forward (:557-563)"""
    x1 = F.prelu(F.conv_transpose2d(x1, sd[p + ".deconv.0.weight"], sd[p + ".deconv.0.bias"], 2, 1), sd[p + ".deconv.1.weight"])
    y = _cp(sd, p + ".conv2", _cp(sd, p + ".conv1", torch.cat((x1, x2), 1)))
    return cbam(sd, p + ".cbam", y) if p + ".cbam.ChannelGate.mlp.1.weight" in sd else y


def feature_net(sd, x, level, p="flownet.fmap"):
    """FeatureNet.forward (:582-597): 5-level U-Net, decoded back to 1/16 (level 0), 1/8 (1) or 1/4 (2)"""
    if x.shape[1] != 17:
        x = _cp(sd, p + ".conv0", x, 1, 1)
    x2 = _unet_conv(sd, p + ".conv1", x)
    x4 = _unet_conv(sd, p + ".conv2", x2)
    x8 = _unet_conv(sd, p + ".conv3", x4)
    x16 = _unet_conv(sd, p + ".conv4", x8)
    x32 = _unet_conv(sd, p + ".conv5", x16)
    y = _up_conv(sd, p + ".deconv5", x32, x16)
    if level != 0:
        y = _up_conv(sd, p + ".deconv4", y, x8)
        if level == 2:
            y = _up_conv(sd, p + ".deconv3", y, x4)
    return y


def if_block(sd, p, x, scale, level):
    """IFBlock.forward (:640-651): residual conv stack, 4-channel flow, convex up-sampling by `level`, resize by `scale`"""
    y = x
    for i in range(6):
        y = _cp(sd, f"{p}.convblock.{i}", y)
    x = y + x
    flow = F.conv2d(x, sd[p + ".flowconv.weight"], sd[p + ".flowconv.bias"], 1, 1)
    mask = F.conv2d(x, sd[f"{p}.maskconvx{level}.weight"], sd[f"{p}.maskconvx{level}.bias"])
    n, _, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, level, level, h, w), dim=2)
    up = F.unfold(level * flow, [3, 3], padding=1).view(n, 4, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(n, 4, level * h, level * w)
    return F.interpolate(up, scale_factor=scale, mode="bilinear", align_corners=False) * scale


def ifunet(sd, x, scale=1.0, timestep=0.5, ensemble=True):
    """IFUNet.forward (:663-743) -> flow [N,4,H,W], warped_img0, warped_img1"""
    c = x.shape[1] // 2
    img0, img1 = x[:, :c], x[:, c:]
    t = (x[:, :1].clone() * 0 + 1) * timestep
    w0, w1 = img0, img1
    flow = None
    levels = (16, 8, 4)

    def estimate(i, parts, prev):
        inp = torch.cat(parts, 1)
        ftmp = prev
        if scale != 1:
            inp = F.interpolate(inp, scale_factor=scale, mode="bilinear", align_corners=False)
            if prev is not None:
                ftmp = F.interpolate(prev, scale_factor=scale, mode="bilinear", align_corners=False) * scale
        if prev is not None:
            inp = torch.cat((inp, ftmp), 1)
        return if_block(sd, f"flownet.block{i}", feature_net(sd, inp, i), 1.0 / scale, levels[i])

    for i in range(3):
        if flow is not None:
            flow = flow + estimate(i, (img0, img1, t, w0, w1), flow)
            if ensemble:
                flow2 = flow + estimate(i, (img1, img0, 1 - t, w0, w1), flow)
                flow = (flow + flow2) / 2
        else:
            flow = estimate(i, (img0, img1, t), None)
            if ensemble:
                flow2 = estimate(i, (img1, img0, 1 - t), None)
                flow = (flow + flow2) / 2
        w0, w1 = warp(img0, flow[:, :2]), warp(img1, flow[:, 2:4])
    return flow, w0, w1


def rrdbnet(sd, img0, img1, w0, w1, flow, p="fusionnet", n_blocks=6):
    """RRDBNet.forward (:306-328): blend mask from the quarter-resolution inputs"""
    x = F.interpolate(torch.cat((img0, img1, w0, w1), 1), scale_factor=0.25, mode="bilinear", align_corners=False)
    fl = F.interpolate(flow, scale_factor=0.25, mode="bilinear", align_corners=False) * 0.25
    feat = F.conv2d(torch.cat((x, fl), 1), sd[p + ".conv_first.weight"], sd[p + ".conv_first.bias"], 1, 1)
    body = feat
    for b in range(n_blocks):
        rin = body
        for r in (1, 2, 3):
            q = f"{p}.body.{b}.rdb{r}."
            xs = [body]
            for k in range(1, 5):
                xs.append(F.leaky_relu(F.conv2d(torch.cat(xs, 1), sd[q + f"conv{k}.weight"], sd[q + f"conv{k}.bias"], 1, 1), 0.2))
            body = F.conv2d(torch.cat(xs, 1), sd[q + "conv5.weight"], sd[q + "conv5.bias"], 1, 1) * 0.2 + body
        body = body * 0.2 + rin
    feat = feat + F.conv2d(body, sd[p + ".conv_body.weight"], sd[p + ".conv_body.bias"], 1, 1)
    for name in ("conv_up1", "conv_up2"):
        feat = F.leaky_relu(F.conv2d(F.interpolate(feat, scale_factor=2.0, mode="nearest"), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], 1, 1), 0.2)
    feat = F.leaky_relu(F.conv2d(feat, sd[p + ".conv_hr.weight"], sd[p + ".conv_hr.bias"], 1, 1), 0.2)
    return torch.sigmoid(F.conv2d(feat, sd[p + ".conv_last.weight"], sd[p + ".conv_last.bias"], 1, 1))


def _flow_block(sd, p, x, flow, scale):
    """FlowBlock.forward (:93-114): BatchNorm conv stack at 1/8 of the resized input, 2-channel flow + mask"""
    x = F.interpolate(x, scale_factor=1.0 / scale, mode="bilinear", align_corners=False)
    if flow is not None:
        flow = F.interpolate(flow, scale_factor=1.0 / scale, mode="bilinear", align_corners=False) * 1.0 / scale
        x = torch.cat((x, flow), 1)
    feat = x
    for i in range(3):
        feat = _cbp(sd, f"{p}.conv0.{i}", feat, 2)
    y = feat
    for i in range(6):
        y = _cbp(sd, f"{p}.convblock.{i}", y)
    feat = y + feat
    tmp = F.conv_transpose2d(feat, sd[p + ".lastconv.weight"], sd[p + ".lastconv.bias"], 2, 1)
    tmp = F.interpolate(tmp, scale_factor=scale * 4, mode="bilinear", align_corners=False)
    return tmp[:, :2] * scale * 4, tmp[:, 2:3]


def resynnet(sd, x, deg, scale=(4, 2, 1), p="refinenet"):
    """ResynNet.forward with training=False, blend=True (:163-192): each input image is aligned to the merged frame `deg` and
    refined; the results and `deg` are blended with a softmax over their (clamped) masks"""
    masks, imgs = [], []
    for i in range(x.shape[1] // 3):
        img = x[:, i * 3:i * 3 + 3]
        flow = mask = None
        for b in range(3):
            if flow is not None:
                fd, md = _flow_block(sd, f"{p}.block{b}", torch.cat((img, deg, wimg, mask), 1), flow, scale[b])
                flow, mask = flow + fd, mask + md
            else:
                flow, mask = _flow_block(sd, f"{p}.block{b}", torch.cat((img, deg), 1), None, scale[b])
            wimg = warp(img, flow)
        fdown = F.interpolate(flow, scale_factor=0.25, mode="bilinear", align_corners=False) * 0.25
        c0 = warp(_cp(sd, p + ".context0.1", _cp(sd, p + ".context0.0", img, 2), 2), fdown)
        c1 = _cp(sd, p + ".context1.1", _cp(sd, p + ".context1.0", wimg, 2), 2)
        d = F.conv_transpose2d(torch.cat((c0, c1), 1), sd[p + ".decode.0.weight"], sd[p + ".decode.0.bias"], 2, 1)
        d = torch.tanh(F.conv_transpose2d(d, sd[p + ".decode.1.weight"], sd[p + ".decode.1.bias"], 2, 1))
        masks.append(mask)
        imgs.append(torch.clamp(wimg + d, 0, 1))
    masks.append(mask * 0)
    imgs.append(deg)
    m = F.softmax(torch.clamp(torch.cat(masks, 1), -4, 4), dim=1)
    merged = 0
    for i, im in enumerate(imgs):
        merged += im * m[:, i:i + 1]
    return merged


def ifunet_forward(sd, img0, img1, timestep=0.5, scale=1.0, ensemble=False):
    """IFUNetModel.forward (:753-766)"""
    n, c, h, w = img0.shape
    ph, pw = ((h - 1) // 64 + 1) * 64, ((w - 1) // 64 + 1) * 64
    img0, img1 = F.pad(img0, (0, pw - w, 0, ph - h)), F.pad(img1, (0, pw - w, 0, ph - h))
    imgs = torch.cat((img0, img1), 1)
    flow, w0, w1 = ifunet(sd, imgs, scale, timestep, ensemble)
    mask = rrdbnet(sd, img0, img1, w0, w1, flow)
    merged = w0 * mask + w1 * (1 - mask)
    return resynnet(sd, imgs, merged)[:, :, :h, :w]


def ifunet_vfi(sd, frames, multiplier=2, scale_factor=1.0, ensemble=True, states=None):
    """Node-level oracle (ifunet/__init__.py:32-59 + generic_frame_loop, int multiplier)"""
    x = frames[..., :3].permute(0, 3, 1, 2).float()
    out = []
    with torch.inference_mode():
        for i in range(len(x) - 1):
            out.append(x[i:i + 1])
            if states is not None and states.is_frame_skipped(i):
                continue
            for k in range(1, multiplier):
                out.append(ifunet_forward(sd, x[i:i + 1], x[i + 1:i + 2], k / multiplier, scale_factor, ensemble))
        out.append(x[-1:])
    return torch.cat(out, 0).permute(0, 2, 3, 1).contiguous()
