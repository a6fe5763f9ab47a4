"""ORACLE — test infrastructure only.  Never imported by the product path.

CPU restatement (torch-CPU fp32, functional style, no nn.Module) of the reference's
RIFE "4.7" path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file; the shipped node (comfyui-frame-interpolation_amd/rife.py) fails
loudly when the HIP library is missing instead of falling back to this.

What is restated (reference file:line under /root/reference):
  * IFNet.forward, arch "4.7" live path          vfi_models/rife/rife_arch.py:465-732
  * IFBlock.forward                              vfi_models/rife/rife_arch.py:237-276
  * ResConv.forward                              vfi_models/rife/rife_arch.py:20-28
  * warp                                         vfi_models/rife/rife_arch.py:31-70
  * RIFE_VFI.vfi scheduling / output interleave  vfi_models/rife/__init__.py:146-239
  * preprocess_frames / postprocess_frames       vfi_utils.py:139-143

The arithmetic of conv2d / conv_transpose2d / grid_sample / interpolate is torch's own
(the reference's third-party dependency, unpinned in requirements-no-cupy.txt:1); the
oracle is pinned de facto to this image's torch 2.10.0 CPU build, exactly like the
reference's CPU path.

Pinning: oracle/validate_vs_reference.py runs the real reference modules (imported from
/root/reference behind oracle/stubs) against this file — bit-exact agreement is required —
and oracle/make_golden.py writes reference outputs to tests/golden/*.npz, which
tests/test_oracle_golden.py re-checks wherever /root/reference is absent.
"""
import torch
import torch.nn.functional as F

_grid_cache = {}


def warp(ten_input, ten_flow):
    """rife_arch.py:31-70 — backward bilinear warp, border padding, align_corners=True."""
    b, _, h, w = ten_flow.shape
    k = (b, h, w)
    if k not in _grid_cache:
        hor = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, -1, h, -1)
        ver = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, -1, -1, w)
        _grid_cache[k] = torch.cat([hor, ver], 1)
    flow = torch.cat(
        [
            ten_flow[:, 0:1] / ((ten_input.shape[3] - 1.0) / 2.0),
            ten_flow[:, 1:2] / ((ten_input.shape[2] - 1.0) / 2.0),
        ],
        1,
    )
    g = (_grid_cache[k] + flow).permute(0, 2, 3, 1)
    return F.grid_sample(ten_input, g, mode="bilinear", padding_mode="border", align_corners=True)


def ifblock(sd, prefix, x, flow, scale, with_feat=False):
    """rife_arch.py:237-276 for arch 4.7 / 4.17 (conv0 x2, 8 ResConv, deconv + PixelShuffle); with_feat: arch 4.26, whose
    lastconv has 4*13 channels and which also returns tmp[:, 5:] (:267-273)."""
    x = F.interpolate(x, scale_factor=1.0 / scale, mode="bilinear", align_corners=False)
    if flow is not None:
        flow = F.interpolate(flow, scale_factor=1.0 / scale, mode="bilinear", align_corners=False) * 1.0 / scale
        x = torch.cat((x, flow), 1)
    p = prefix
    feat = F.leaky_relu(F.conv2d(x, sd[p + "conv0.0.0.weight"], sd[p + "conv0.0.0.bias"], 2, 1), 0.2)
    feat = F.leaky_relu(F.conv2d(feat, sd[p + "conv0.1.0.weight"], sd[p + "conv0.1.0.bias"], 2, 1), 0.2)
    for i in range(8):
        q = p + f"convblock.{i}."
        y = F.conv2d(feat, sd[q + "conv.weight"], sd[q + "conv.bias"], 1, 1)
        feat = F.leaky_relu(y * sd[q + "beta"] + feat, 0.2)
    tmp = F.conv_transpose2d(feat, sd[p + "lastconv.0.weight"], sd[p + "lastconv.0.bias"], 2, 1)
    tmp = F.pixel_shuffle(tmp, 2)
    tmp = F.interpolate(tmp, scale_factor=scale, mode="bilinear", align_corners=False)
    if with_feat:
        return tmp[:, :4] * scale, tmp[:, 4:5], tmp[:, 5:]
    return tmp[:, :4] * scale, tmp[:, 4:5]


def encode(sd, img):
    """rife_arch.py:414-416 — Conv2d(3,16,3,2,1) -> ConvTranspose2d(16,4,4,2,1), no activation."""
    e = F.conv2d(img, sd["encode.0.weight"], sd["encode.0.bias"], 2, 1)
    return F.conv_transpose2d(e, sd["encode.1.weight"], sd["encode.1.bias"], 2, 1)


def encode417(sd, img):
    """Head_417.forward, rife_arch.py:355-375: Conv(3,32,s2) lrelu Conv(32,32) lrelu Conv(32,32) lrelu Deconv(32,8)"""
    x = F.leaky_relu(F.conv2d(img, sd["encode.cnn0.weight"], sd["encode.cnn0.bias"], 2, 1), 0.2)
    x = F.leaky_relu(F.conv2d(x, sd["encode.cnn1.weight"], sd["encode.cnn1.bias"], 1, 1), 0.2)
    x = F.leaky_relu(F.conv2d(x, sd["encode.cnn2.weight"], sd["encode.cnn2.bias"], 1, 1), 0.2)
    return F.conv_transpose2d(x, sd["encode.cnn3.weight"], sd["encode.cnn3.bias"], 2, 1)


def ifnet47_forward(sd, img0, img1, timestep, scale_list=(8, 4, 2, 1), return_aux=False, arch="4.7"):
    """rife_arch.py:465-732, arch "4.7" / "4.17" (same live path :501-503,543-548,629-645,698-705 — they differ in the
    encoder and the channel counts only), ensemble=False (the only path the node reaches, App. C1).

    img0/img1: [B,3,H,W] f32;  timestep: [B,1,1,1] tensor.  Returns [B,3,H,W]."""
    if arch == "4.26":
        return ifnet426_forward(sd, img0, img1, timestep, scale_list, return_aux)
    enc = {"4.7": encode, "4.17": encode417}[arch]
    img0 = torch.clamp(img0, 0, 1)
    img1 = torch.clamp(img1, 0, 1)
    n, c, h, w = img0.shape
    ph = ((h - 1) // 64 + 1) * 64
    pw = ((w - 1) // 64 + 1) * 64
    padding = (0, pw - w, 0, ph - h)
    img0 = F.pad(img0, padding)
    img1 = F.pad(img1, padding)
    timestep = timestep.repeat(1, 1, img0.shape[2], img0.shape[3])
    f0 = enc(sd, img0[:, :3])
    f1 = enc(sd, img1[:, :3])
    warped_img0, warped_img1 = img0, img1
    flow = None
    mask = None
    aux = []
    for i in range(4):
        p = f"block{i}."
        if flow is None:
            flow, mask = ifblock(sd, p, torch.cat((img0[:, :3], img1[:, :3], f0, f1, timestep), 1), None, scale_list[i])
        else:
            fd, m0 = ifblock(
                sd,
                p,
                torch.cat(
                    (warped_img0[:, :3], warped_img1[:, :3], warp(f0, flow[:, :2]), warp(f1, flow[:, 2:4]), timestep, mask),
                    1,
                ),
                flow,
                scale_list[i],
            )
            flow = flow + fd
            mask = m0
        warped_img0 = warp(img0, flow[:, :2])
        warped_img1 = warp(img1, flow[:, 2:4])
        if return_aux:
            aux.append((flow.clone(), mask.clone()))
    mask = torch.sigmoid(mask)
    merged = warped_img0 * mask + warped_img1 * (1 - mask)
    out = merged[:, :, :h, :w]
    return (out, aux) if return_aux else out


def ifnet426_forward(sd, img0, img1, timestep, scale_list=(16, 8, 4, 2, 1), return_aux=False):
    """rife_arch.py:465-732, arch "4.26" (:451-457 five blocks, Head encoder :378-398; live path :501-503,512-526,
    :555-583,698-705): every block also returns 8 feature channels that are appended to the next block's input."""
    img0 = torch.clamp(img0, 0, 1)
    img1 = torch.clamp(img1, 0, 1)
    n, c, h, w = img0.shape
    ph = ((h - 1) // 64 + 1) * 64
    pw = ((w - 1) // 64 + 1) * 64
    img0 = F.pad(img0, (0, pw - w, 0, ph - h))
    img1 = F.pad(img1, (0, pw - w, 0, ph - h))
    timestep = timestep.repeat(1, 1, img0.shape[2], img0.shape[3])
    f0 = encode417(sd, img0[:, :3])   # Head has Head_417's structure (16 / 4 channels instead of 32 / 8)
    f1 = encode417(sd, img1[:, :3])
    warped_img0, warped_img1 = img0, img1
    flow = mask = feat = None
    aux = []
    for i in range(5):
        p = f"block{i}."
        if flow is None:
            flow, mask, feat = ifblock(sd, p, torch.cat((img0[:, :3], img1[:, :3], f0, f1, timestep), 1), None, scale_list[i], True)
        else:
            x = torch.cat((warped_img0[:, :3], warped_img1[:, :3], warp(f0, flow[:, :2]), warp(f1, flow[:, 2:4]), timestep, mask, feat), 1)
            fd, m0, feat = ifblock(sd, p, x, flow, scale_list[i], True)
            flow = flow + fd
            mask = m0
        warped_img0 = warp(img0, flow[:, :2])
        warped_img1 = warp(img1, flow[:, 2:4])
        if return_aux:
            aux.append((flow.clone(), mask.clone(), feat.clone()))
    mask = torch.sigmoid(mask)
    out = (warped_img0 * mask + warped_img1 * (1 - mask))[:, :, :h, :w]
    return (out, aux) if return_aux else out


# ---------------------------------------------------------------------------------------------
# arch "4.0" (sudo_rife4_269.662_testV1_scale1.pth): PReLU convs, no encoder, flow/mask ACCUMULATED over the blocks,
# optional Contextnet + Unet refinement.  Here the node's two booleans finally matter, under crossed names: the node passes
# (fast_mode, ensemble) positionally into forward(..., training, fastmode) (rife/__init__.py:200-206, rife_arch.py:465-475).
# ---------------------------------------------------------------------------------------------
def _cp(sd, p, x, stride=1):
    """conv() for arch 4.0: Conv2d(3, stride, 1) + PReLU(out_planes), rife_arch.py:82-94"""
    return F.prelu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride, 1), sd[p + ".1.weight"])


def ifblock40(sd, p, x, flow, scale):
    """IFBlock.forward for arch 4.0, rife_arch.py:237-261: conv0 x2 (stride 2), convblock(feat) + feat, ConvTranspose2d(c,5),
    up-resize by 2*scale"""
    x = F.interpolate(x, scale_factor=1.0 / scale, mode="bilinear", align_corners=False)
    if flow is not None:
        flow = F.interpolate(flow, scale_factor=1.0 / scale, mode="bilinear", align_corners=False) * 1.0 / scale
        x = torch.cat((x, flow), 1)
    feat = _cp(sd, p + "conv0.1", _cp(sd, p + "conv0.0", x, 2), 2)
    y = feat
    for i in range(8):
        y = _cp(sd, p + f"convblock.{i}", y)
    feat = y + feat
    tmp = F.conv_transpose2d(feat, sd[p + "lastconv.weight"], sd[p + "lastconv.bias"], 2, 1)
    tmp = F.interpolate(tmp, scale_factor=scale * 2, mode="bilinear", align_corners=False)
    return tmp[:, :4] * scale * 2, tmp[:, 4:5]


def _conv2_40(sd, p, x):
    return _cp(sd, p + ".conv2", _cp(sd, p + ".conv1", x, 2))


def contextnet40(sd, x, flow):
    """Contextnet.forward, rife_arch.py:288-313"""
    out = []
    for i in range(1, 5):
        x = _conv2_40(sd, f"contextnet.conv{i}", x)
        flow = F.interpolate(flow, scale_factor=0.5, mode="bilinear", align_corners=False) * 0.5
        out.append(warp(x, flow))
    return out


def unet40(sd, img0, img1, w0, w1, mask, flow, c0, c1):
    """Unet.forward, rife_arch.py:330-342"""
    def dec(p, x):
        return F.prelu(F.conv_transpose2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 2, 1), sd[p + ".1.weight"])

    s0 = _conv2_40(sd, "unet.down0", torch.cat((img0, img1, w0, w1, mask, flow), 1))
    s1 = _conv2_40(sd, "unet.down1", torch.cat((s0, c0[0], c1[0]), 1))
    s2 = _conv2_40(sd, "unet.down2", torch.cat((s1, c0[1], c1[1]), 1))
    s3 = _conv2_40(sd, "unet.down3", torch.cat((s2, c0[2], c1[2]), 1))
    x = dec("unet.up0", torch.cat((s3, c0[3], c1[3]), 1))
    x = dec("unet.up1", torch.cat((x, s2), 1))
    x = dec("unet.up2", torch.cat((x, s1), 1))
    x = dec("unet.up3", torch.cat((x, s0), 1))
    return torch.sigmoid(F.conv2d(x, sd["unet.conv.weight"], sd["unet.conv.bias"], 1, 1))


def ifnet40_forward(sd, img0, img1, timestep, scale_list, training=True, fastmode=True, return_aux=False):
    """rife_arch.py:465-732, arch "4.0", ensemble=False.  ``scale_list`` is a LIST and is doubled IN PLACE when block 1
    reports flows above 32 px in both directions and ``training`` is False (:598-607) — the node builds the list once per
    call and passes the same object to every batch (rife/__init__.py:157-160,200-206), so the doubling sticks."""
    img0 = torch.clamp(img0, 0, 1)
    img1 = torch.clamp(img1, 0, 1)
    n, c, h, w = img0.shape
    ph = ((h - 1) // 64 + 1) * 64
    pw = ((w - 1) // 64 + 1) * 64
    img0 = F.pad(img0, (0, pw - w, 0, ph - h))
    img1 = F.pad(img1, (0, pw - w, 0, ph - h))
    timestep = timestep.repeat(1, 1, img0.shape[2], img0.shape[3])
    warped_img0, warped_img1 = img0, img1
    flow = mask = None
    aux = []
    for i in range(4):
        p = f"block{i}."
        if flow is None:
            flow, mask = ifblock40(sd, p, torch.cat((img0[:, :3], img1[:, :3], timestep), 1), None, scale_list[i])
        else:
            f0, m0 = ifblock40(sd, p, torch.cat((warped_img0[:, :3], warped_img1[:, :3], timestep, mask), 1), flow, scale_list[i])
            if i == 1 and f0[:, :2].abs().max() > 32 and f0[:, 2:4].abs().max() > 32 and not training:
                for k in range(4):
                    scale_list[k] *= 2
                flow, mask = ifblock40(sd, "block0.", torch.cat((img0[:, :3], img1[:, :3], timestep), 1), None, scale_list[0])
                warped_img0 = warp(img0, flow[:, :2])
                warped_img1 = warp(img1, flow[:, 2:4])
                f0, m0 = ifblock40(sd, p, torch.cat((warped_img0[:, :3], warped_img1[:, :3], timestep, mask), 1), flow, scale_list[i])
            flow = flow + f0
            mask = mask + m0
        warped_img0 = warp(img0, flow[:, :2])
        warped_img1 = warp(img1, flow[:, 2:4])
        if return_aux:
            aux.append((flow.clone(), mask.clone()))
    sm = torch.sigmoid(mask)
    merged = warped_img0 * sm + warped_img1 * (1 - sm)
    if not fastmode:
        c0 = contextnet40(sd, img0, flow[:, :2])
        c1 = contextnet40(sd, img1, flow[:, 2:4])
        tmp = unet40(sd, img0, img1, warped_img0, warped_img1, mask, flow, c0, c1)
        merged = torch.clamp(merged + (tmp[:, :3] * 2 - 1), 0, 1)
    out = merged[:, :, :h, :w]
    return (out, aux) if return_aux else out


# ---------------------------------------------------------------------------------------------
# node level (vfi_models/rife/__init__.py:146-239)
# ---------------------------------------------------------------------------------------------

def rife_tasks(n_frames, multiplier, states=None):
    """rife/__init__.py:149-174: per-pair multipliers (list padded with 2) and the flat task list."""
    n_pairs = n_frames - 1
    if isinstance(multiplier, int):
        multipliers = [int(multiplier)] * n_pairs
    else:
        multipliers = list(map(int, multiplier))
        multipliers += [2] * (n_pairs - len(multipliers))
    tasks = []
    for pair_idx in range(n_pairs):
        if states is not None and states.is_frame_skipped(pair_idx):
            continue
        m = multipliers[pair_idx]
        for step in range(1, m):
            tasks.append((pair_idx, step / m))
    return multipliers, tasks


def rife_vfi(sd, frames, multiplier=2, scale_factor=1.0, batch_size=1, states=None, arch="4.7", fast_mode=False, ensemble=False):
    """Whole-node oracle: frames [N,H,W,C] f32 CPU -> [N_out,H,W,3] f32 CPU."""
    x = frames[..., :3].permute(0, 3, 1, 2)  # preprocess_frames, vfi_utils.py:139-140
    n_pairs = len(x) - 1
    _, tasks = rife_tasks(len(x), multiplier, states)
    scale_list = [8 / scale_factor, 4 / scale_factor, 2 / scale_factor, 1 / scale_factor]
    if arch == "4.26":   # rife/__init__.py:155-156
        scale_list = [16 / scale_factor] + scale_list
    results = {i: [] for i in range(n_pairs)}
    pos = 0
    with torch.inference_mode():
        while pos < len(tasks):
            bt = tasks[pos : pos + batch_size]
            f0 = torch.cat([x[p : p + 1] for p, _ in bt], 0).to(torch.float32)
            f1 = torch.cat([x[p + 1 : p + 2] for p, _ in bt], 0).to(torch.float32)
            ts = torch.tensor([t for _, t in bt], dtype=torch.float32).view(-1, 1, 1, 1)
            if arch == "4.0":   # the two booleans land on forward()'s `training` / `fastmode` (rife/__init__.py:200-206)
                mid = ifnet40_forward(sd, f0, f1, ts, scale_list, fast_mode, ensemble).clamp(0, 1)
            else:
                mid = ifnet47_forward(sd, f0, f1, ts, scale_list, arch=arch).clamp(0, 1)
            for i, (p, _) in enumerate(bt):
                results[p].append(mid[i : i + 1])
            pos += len(bt)
    out = []
    for p in range(n_pairs):
        out.append(x[p : p + 1])
        out.extend(results[p])
    out.append(x[-1:])
    return torch.cat(out, 0).to(torch.float32).permute(0, 2, 3, 1)[..., :3].contiguous()
