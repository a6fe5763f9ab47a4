"""ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the product path
never imports this module.

CPU restatement (torch, fp32) of the reference's IFRNet path:
    vfi_models/ifrnet/IFRNet_L_arch.py / IFRNet_S_arch.py   (IRFNet_L / IRFNet_S .forward, warp, resize, ResBlock)
    vfi_models/ifrnet/__init__.py:32-57                      (the node's call into generic_frame_loop)
    vfi_utils.py:149-389                                      (generic_frame_loop, timestep mode)
written as plain functions over the checkpoint's state_dict.  Pinned bit-exactly against the reference modules and the
reference node on seeded weights by oracle/validate_ifrnet_vs_reference.py (oracle/VALIDATION_IFRNET.log); golden
vectors of the reference itself in tests/golden/ifrnet_*.npz.

Reference behaviour restated as is: the node calls ``model(frame_0, frame_1, timestep, scale_factor)`` while the
signature is ``forward(img0, img1, scale_factor=1.0, timestep=0.5)`` (ifrnet/__init__.py:49-50 vs IFRNet_L_arch.py:225),
so the loop's TIMESTEP k/multiplier is the network's working-resolution factor and the node's ``scale_factor`` widget is
the time embedding.  ``ifrnet_vfi`` below reproduces exactly that.
"""
import torch
import torch.nn.functional as F


def warp(img, flow):
    """IFRNet_L_arch.py:9-29"""
    B, _, H, W = flow.shape
    xx = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W).expand(B, -1, H, -1)
    yy = torch.linspace(-1.0, 1.0, H).view(1, 1, H, 1).expand(B, -1, -1, W)
    grid = torch.cat([xx, yy], 1).to(img)
    flow_ = torch.cat([flow[:, 0:1] / ((W - 1.0) / 2.0), flow[:, 1:2] / ((H - 1.0) / 2.0)], 1)
    grid_ = (grid + flow_).permute(0, 2, 3, 1)
    return F.grid_sample(input=img, grid=grid_, mode="bilinear", padding_mode="border", align_corners=True)


def resize(x, scale_factor):
    """IFRNet_L_arch.py:38-41"""
    return F.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=False)


def _convrelu(sd, p, x, stride=1):
    w = sd[p + ".0.weight"]
    x = F.conv2d(x, w, sd[p + ".0.bias"], stride, w.shape[-1] // 2)
    return F.prelu(x, sd[p + ".1.weight"])


def _resblock(sd, p, x):
    """ResBlock.forward (IFRNet_L_arch.py:113-123): the convs on the last ``side`` channels overwrite them in place."""
    side = sd[p + ".conv2.0.weight"].shape[0]
    out = _convrelu(sd, p + ".conv1", x)
    out = torch.cat([out[:, :-side], _convrelu(sd, p + ".conv2", out[:, -side:])], 1)
    out = _convrelu(sd, p + ".conv3", out)
    out = torch.cat([out[:, :-side], _convrelu(sd, p + ".conv4", out[:, -side:])], 1)
    out = F.conv2d(out, sd[p + ".conv5.weight"], sd[p + ".conv5.bias"], 1, 1)
    return F.prelu(x + out, sd[p + ".prelu.weight"])


def _decoder(sd, d, f_in):
    p = f"decoder{d}.convblock"
    x = _convrelu(sd, p + ".0", f_in)
    x = _resblock(sd, p + ".1", x)
    return F.conv_transpose2d(x, sd[p + ".2.weight"], sd[p + ".2.bias"], 2, 1)


def encoder(sd, img):
    """Encoder.forward (IFRNet_L_arch.py:126-147)"""
    fs = []
    x = img
    for lvl in range(1, 5):
        x = _convrelu(sd, f"encoder.pyramid{lvl}.0", x, 2)
        x = _convrelu(sd, f"encoder.pyramid{lvl}.1", x)
        fs.append(x)
    return fs


def ifrnet_forward(sd, img0, img1, scale_factor=1.0, timestep=0.5, return_aux=False):
    """IRFNet_L.forward / IRFNet_S.forward (IFRNet_L_arch.py:225-293); img [N,3,H,W] fp32."""
    n, c, h, w = img0.shape
    ph = ((h - 1) // 64 + 1) * 64
    pw = ((w - 1) // 64 + 1) * 64
    img0 = F.pad(img0, (0, pw - w, 0, ph - h))
    img1 = F.pad(img1, (0, pw - w, 0, ph - h))
    embt = torch.tensor([timestep] * n).view(n, 1, 1, 1).float()
    mean_ = torch.cat([img0, img1], 2).mean(1, keepdim=True).mean(2, keepdim=True).mean(3, keepdim=True)
    img0 = img0 - mean_
    img1 = img1 - mean_
    img0_ = resize(img0, scale_factor)
    img1_ = resize(img1, scale_factor)
    f0 = encoder(sd, img0_)
    f1 = encoder(sd, img1_)
    b, _, hh, ww = f0[3].shape
    out4 = _decoder(sd, 4, torch.cat([f0[3], f1[3], embt.repeat(1, 1, hh, ww)], 1))
    up0, up1, ft = out4[:, 0:2], out4[:, 2:4], out4[:, 4:]
    flows = [(up0, up1)]
    for d, lvl in ((3, 2), (2, 1), (1, 0)):
        f_in = torch.cat([ft, warp(f0[lvl], up0), warp(f1[lvl], up1), up0, up1], 1)
        out = _decoder(sd, d, f_in)
        up0 = out[:, 0:2] + 2.0 * resize(up0, 2.0)
        up1 = out[:, 2:4] + 2.0 * resize(up1, 2.0)
        ft = out[:, 4:]
        flows.append((up0, up1))
    mask = torch.sigmoid(out[:, 4:5])
    res = out[:, 5:]
    up0 = resize(up0, 1.0 / scale_factor) * (1.0 / scale_factor)
    up1 = resize(up1, 1.0 / scale_factor) * (1.0 / scale_factor)
    mask = resize(mask, 1.0 / scale_factor)
    res = resize(res, 1.0 / scale_factor)
    merged = mask * warp(img0, up0) + (1 - mask) * warp(img1, up1) + mean_
    pred = torch.clamp(merged + res, 0, 1)[:, :, :h, :w]
    if return_aux:
        return pred, {"flows": flows, "final_flow": (up0, up1), "mean": mean_}
    return pred


def _loop(sd, x, multiplier, scale_factor, states):
    """_generic_frame_loop, timestep mode, batch_size 1 (vfi_utils.py:149-338) with the node's positional call."""
    out = []
    for i in range(len(x) - 1):
        out.append(x[i:i + 1])
        if states is not None and states.is_frame_skipped(i):
            continue
        for k in range(1, multiplier):
            timestep = k / multiplier
            # return_middle_frame(frame_0, frame_1, timestep, model, scale_factor) -> model(frame_0, frame_1, timestep, scale_factor)
            out.append(ifrnet_forward(sd, x[i:i + 1], x[i + 1:i + 2], timestep, scale_factor))
    out.append(x[-1:])
    return out


def ifrnet_vfi(sd, frames, multiplier=2, scale_factor=1.0, states=None):
    """Node-level oracle (ifrnet/__init__.py:32-57 + generic_frame_loop, vfi_utils.py:339-389); frames [N,H,W,C] fp32."""
    x = frames[..., :3].permute(0, 3, 1, 2).float()
    with torch.inference_mode():
        if type(multiplier) == int:
            out = _loop(sd, x, multiplier, scale_factor, states)
        else:
            ms = list(map(int, multiplier))
            ms += [2] * (len(x) - len(ms) - 1)
            out = []
            for i in range(len(x) - 1):
                if ms[i] == 0:
                    continue
                part = _loop(sd, x[i:i + 2], ms[i], scale_factor, states)
                out.extend(part if i == len(x) - 2 else part[:-1])
    return torch.cat(out, 0).permute(0, 2, 3, 1).contiguous()
