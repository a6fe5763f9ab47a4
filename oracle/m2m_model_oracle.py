"""ORACLE — test infrastructure only.  Never imported by the product path.

CPU restatement (torch-CPU fp32, functional) of the reference's M2M model, vfi_models/m2m/M2M_arch.py, with the two
custom CUDA ops replaced by the plain-C restatements of their kernel text (oracle/m2m_ops.c).

Pinning: oracle/validate_m2m_vs_reference.py imports the reference's M2M_arch.py with a stand-in ``vfi_models.ops``
module that forwards to the same C restatements and requires bit-exact agreement of everything else (convs,
warps, resizes, statistics, splat pre/post-processing).  The two ops themselves remain UNPINNED by execution
(no CPU path exists in the reference).

Restated (file:line in vfi_models/m2m/M2M_arch.py): backwarp :24-92; Basic DSL (evenize/sconv/conv/prelu) :100-411;
Network.Extractor / Decoder / bidir :415-546; forwarp_mframe_mask :551-581; conv/deconv/Conv2/ImgPyramid :589-663;
EncDec :665-848; M2M_PWC.forward :894-1037; node loop vfi_utils.py:149-389 (generic_frame_loop, timestep mode).
"""
import torch
import torch.nn.functional as F

from . import m2m_oracle


def softsplat(ten_in, ten_flow):
    return torch.from_numpy(m2m_oracle.softsplat_sum(ten_in.numpy(), ten_flow.numpy()))


def costvol(one, two):
    return torch.from_numpy(m2m_oracle.costvol(one.numpy(), two.numpy()))


def backwarp(ten_in, ten_flow):
    """M2M_arch.py:24-92 — bilinear, zeros padding, align_corners=True."""
    _, _, h, w = ten_flow.shape
    hor = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, -1).repeat(1, 1, h, 1)
    ver = torch.linspace(-1.0, 1.0, h).view(1, 1, -1, 1).repeat(1, 1, 1, w)
    grid = torch.cat([hor, ver], 1)
    if w == h:
        ten_flow = ten_flow * (2.0 / (h - 1.0))
    else:
        ten_flow = ten_flow * torch.tensor([2.0 / (w - 1.0), 2.0 / (h - 1.0)]).view(1, 2, 1, 1)
    return F.grid_sample(ten_in, (grid + ten_flow).permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros",
                         align_corners=True)


def _evenize_repl(x):
    pad = [0, 1 if x.shape[3] % 2 else 0, 0, 1 if x.shape[2] % 2 else 0]
    return F.pad(x, pad, mode="replicate") if max(pad) else x


def _conv_repl(sd, key, x):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), sd[key + ".weight"], sd[key + ".bias"])


def _prelu1(sd, key, x):
    return F.prelu(x, sd[key + ".weight"])


def extractor_stage(sd, p, x):
    """Basic("evenize(replpad)-sconv(2)-prelu-conv(3,replpad)-prelu-conv(3,replpad)-prelu") :421-435"""
    x = _evenize_repl(x)
    x = _prelu1(sd, p + ".netMain.1", F.conv2d(x, sd[p + ".netMain.0.weight"], sd[p + ".netMain.0.bias"], stride=2))
    x = _prelu1(sd, p + ".netMain.3", _conv_repl(sd, p + ".netMain.2", x))
    x = _prelu1(sd, p + ".netMain.5", _conv_repl(sd, p + ".netMain.4", x))
    return x


def extractor(sd, x):
    one = extractor_stage(sd, "netFlow.netExtractor.netOne", x)
    two = extractor_stage(sd, "netFlow.netExtractor.netTwo", one)
    thr = extractor_stage(sd, "netFlow.netExtractor.netThr", two)
    fou = F.avg_pool2d(thr, 2, 2, count_include_pad=False)
    fiv = F.avg_pool2d(fou, 2, 2, count_include_pad=False)
    return [one, two, thr, fou, fiv]


def decoder(sd, p, one, two, flow):
    """Network.Decoder.forward :468-503"""
    if flow is not None:
        flow = 2.0 * F.interpolate(flow, scale_factor=2.0, mode="bilinear", align_corners=False)
    main = [one]
    if flow is None:
        main.append(F.prelu(costvol(one, two), sd[p + ".netCostacti.weight"]))
    else:
        main.append(F.prelu(costvol(one, backwarp(two, flow)), sd[p + ".netCostacti.weight"]))
        main.append(flow)
    x = torch.cat(main, 1)
    q = p + ".netMain.netMain."
    for i in range(5):
        x = _prelu1(sd, q + str(2 * i + 1), _conv_repl(sd, q + str(2 * i), x))
    x = _conv_repl(sd, q + "10", x)
    return (flow if flow is not None else 0.0) + x


def bidir(sd, one, two):
    feats = extractor(sd, torch.cat([one, two], 0))
    fo = [f[: one.shape[0]] for f in feats]
    ft = [f[one.shape[0]:] for f in feats]
    names = ["netFlow.netFiv", "netFlow.netFou", "netFlow.netThr", "netFlow.netTwo", "netFlow.netOne"]
    fwd = bwd = None
    for k, n in enumerate(names):
        fwd = decoder(sd, n, fo[-1 - k], ft[-1 - k], fwd)
    for k, n in enumerate(names):
        bwd = decoder(sd, n, ft[-1 - k], fo[-1 - k], bwd)
    return fwd, bwd


def _cp(sd, p, x, stride=1):
    """conv() helper :589-602: Conv2d(k=3,pad=1) + PReLU(per channel)"""
    return F.prelu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride, 1), sd[p + ".1.weight"])


def _conv2(sd, p, x):
    return _cp(sd, p + ".conv2", _cp(sd, p + ".conv1", x, 2))


def _deconv(sd, p, x):
    return F.prelu(F.conv_transpose2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 2, 1), sd[p + ".1.weight"])


def img_pyramid(sd, x):
    p = "MRN.img_pyramid."
    x1 = _conv2(sd, p + "conv1", x)
    x2 = _conv2(sd, p + "conv2", x1)
    x3 = _conv2(sd, p + "conv3", x2)
    x4 = _conv2(sd, p + "conv4", x3)
    return [x1, x2, x3, x4]


def _half(flow):
    return F.interpolate(flow, scale_factor=0.5, mode="bilinear", align_corners=False) * 0.5


def _cube(sd, s3):
    p = "MRN.motion_encdec."
    n = s3.shape[0]
    c = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(s3, 1), sd[p + "conv_C.1.weight"], sd[p + "conv_C.1.bias"])).view(n, 16, -1, 1, 1)
    h = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(s3, (None, 1)), sd[p + "conv_H.1.weight"], sd[p + "conv_H.1.bias"])).view(n, 16, 1, -1, 1)
    w = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(s3, (1, None)), sd[p + "conv_W.1.weight"], sd[p + "conv_W.1.bias"])).view(n, 16, 1, 1, -1)
    return s3 * (c * h * w).mean(1)


def encdec(sd, flow0, flow1, im0, im1, c0, c1):
    """EncDec.forward :718-848"""
    p = "MRN.motion_encdec."
    wim1 = backwarp(im1, flow0)
    wim0 = backwarp(im0, flow1)
    s0 = [_conv2(sd, p + "down0", torch.cat((flow0, im0, wim1), 1))]
    s1 = [_conv2(sd, p + "down0", torch.cat((flow1, im1, wim0), 1))]
    for lvl in range(3):
        flow0, flow1 = _half(flow0), _half(flow1)
        wf0 = backwarp(torch.cat((s0[lvl], c0[lvl]), 1), flow1)
        wf1 = backwarp(torch.cat((s1[lvl], c1[lvl]), 1), flow0)
        s0.append(_conv2(sd, p + f"down{lvl + 1}", torch.cat((s0[lvl], c0[lvl], wf1), 1)))
        s1.append(_conv2(sd, p + f"down{lvl + 1}", torch.cat((s1[lvl], c1[lvl], wf0), 1)))
    s0[3] = _cube(sd, s0[3])
    s1[3] = _cube(sd, s1[3])
    flow0, flow1 = _half(flow0), _half(flow1)
    wf0 = backwarp(torch.cat((s0[3], c0[3]), 1), flow1)
    wf1 = backwarp(torch.cat((s1[3], c1[3]), 1), flow0)
    x0 = _deconv(sd, p + "up0", torch.cat((s0[3], c0[3], wf1), 1))
    x1 = _deconv(sd, p + "up0", torch.cat((s1[3], c1[3], wf0), 1))
    for k, lvl in ((1, 2), (2, 1), (3, 0)):
        x0 = _deconv(sd, p + f"up{k}", torch.cat((s0[lvl], x0), 1))
        x1 = _deconv(sd, p + f"up{k}", torch.cat((s1[lvl], x1), 1))
    m0 = torch.sigmoid(F.conv2d(x0, sd[p + "conv_m.weight"], sd[p + "conv_m.bias"], 1, 1)) * 0.8 + 0.1
    m1 = torch.sigmoid(F.conv2d(x1, sd[p + "conv_m.weight"], sd[p + "conv_m.bias"], 1, 1)) * 0.8 + 0.1
    x0 = F.conv2d(x0, sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 1)
    x1 = F.conv2d(x1, sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 1)
    return x0, x1, m0.repeat(1, 4, 1, 1), m1.repeat(1, 4, 1, 1)


def forwarp_mframe_mask(in1, flow1, t1, in2, flow2, t2, metric1, metric2):
    """:551-581"""
    def one_fdir(ten_in, ten_flow, td, ten_metric):
        e = ten_metric.clip(-20.0, 20.0).exp()
        x = torch.cat([ten_in * td * e, td * e], 1)
        o = softsplat(x.contiguous(), ten_flow.contiguous())
        return o[:, :-1], o[:, -1:] + 0.0000001

    out, norm = 0, 0
    for idx in range(flow1.shape[0]):
        of, nf = one_fdir(in1[idx], flow1[idx], t1[idx], metric1[idx])
        ob, nb = one_fdir(in2[idx], flow2[idx], t2[idx], metric2[idx])
        out += of + ob
        norm += nf + nb
    return out / norm, norm < 0.00001


def m2m_forward(sd, im0, im1, flt_times, ratio=4, return_aux=False):
    """M2M_PWC.forward :894-1037.  im0/im1 [N,3,H,W]; flt_times: list of [N,1,1,1] tensors."""
    branch = 4
    w_, h_ = im0.shape[3], im0.shape[2]
    padr = ((ratio * 16) - (w_ % (ratio * 16))) % (ratio * 16)
    padb = ((ratio * 16) - (h_ % (ratio * 16))) % (ratio * 16)
    im0 = F.pad(im0, [0, padr, 0, padb], mode="replicate")
    im1 = F.pad(im1, [0, padr, 0, padb], mode="replicate")
    N_, C_, H_, W_ = im0.shape
    stats = [im0, im1]
    mean_ = sum([t.mean([1, 2, 3], True) for t in stats]) / len(stats)
    std_ = (sum([t.std([1, 2, 3], False, True).square() + (mean_ - t.mean([1, 2, 3], True)).square() for t in stats]) / len(stats)).sqrt()
    im0_o = (im0 - mean_) / (std_ + 0.0000001)
    im1_o = (im1 - mean_) / (std_ + 0.0000001)
    im0, im1 = im0_o, im1_o
    im0_ = F.interpolate(im0, scale_factor=2.0 / ratio, mode="bilinear", align_corners=False)
    im1_ = F.interpolate(im1, scale_factor=2.0 / ratio, mode="bilinear", align_corners=False)
    fwd, bwd = bidir(sd, im0_, im1_)
    # MotionRefineNet.forward :866-890
    flow0 = ratio * F.interpolate(fwd, scale_factor=ratio, mode="bilinear", align_corners=False)
    flow1 = ratio * F.interpolate(bwd, scale_factor=ratio, mode="bilinear", align_corners=False)
    c0 = img_pyramid(sd, im0)
    c1 = img_pyramid(sd, im1)
    res = encdec(sd, flow0, flow1, im0, im1, c0, c1)
    ten_fwd = flow0.repeat(1, branch, 1, 1) + res[0]
    ten_bwd = flow1.repeat(1, branch, 1, 1) + res[1]
    wei_f, wei_b = res[2], res[3]
    alpha = sd["paramAlpha"]
    outputs = []
    for ft in flt_times:
        i0 = im0_o.repeat(1, branch, 1, 1).reshape(N_ * branch, 3, H_, W_)
        i1 = im1_o.repeat(1, branch, 1, 1).reshape(N_ * branch, 3, H_, W_)
        tf = ten_fwd.reshape(N_ * branch, 2, H_, W_)
        tb = ten_bwd.reshape(N_ * branch, 2, H_, W_)
        wf = wei_f.reshape(N_ * branch, 1, H_, W_)
        wb = wei_b.reshape(N_ * branch, 1, H_, W_)
        t = ft.repeat(1, branch, 1, 1).reshape(N_ * branch, 1, 1, 1)
        photo1 = (1.0 - (wf * (i0 - backwarp(i1, tf)).abs().mean([1], True))).clip(0.001, None).square()
        photo2 = (1.0 - (wb * (i1 - backwarp(i0, tb)).abs().mean([1], True))).clip(0.001, None).square()
        t0 = t
        fl0 = tf * t0
        m0 = alpha * photo1
        t1 = 1.0 - t
        fl1 = tb * t1
        m1 = alpha * photo2
        rs = lambda x, c: x.reshape(N_, branch, c, *x.shape[2:]).permute(1, 0, 2, 3, 4)
        out, mask = forwarp_mframe_mask(rs(i0, 3), rs(fl0, 2), rs(t1, 1), rs(i1, 3), rs(fl1, 2), rs(t0, 1), rs(m0, 1), rs(m1, 1))
        out = out + mask * (rs(t1, 1).mean(0) * im0_o + rs(t0, 1).mean(0) * im1_o)
        outputs.append((out * (std_ + 0.0000001)) + mean_)
    outs = [o[:, :, :h_, :w_] for o in outputs]
    if return_aux:
        return outs, dict(fwd=fwd, bwd=bwd, ten_fwd=ten_fwd, ten_bwd=ten_bwd, wei_f=wei_f, c0=c0, res=res, mean=mean_, std=std_)
    return outs


def _m2m_loop(sd, x, multiplier, states):
    """_generic_frame_loop, timestep mode, batch_size 1 (vfi_utils.py:149-338)"""
    out = []
    for i in range(len(x) - 1):
        out.append(x[i:i + 1])
        if states is not None and states.is_frame_skipped(i):
            continue
        for k in range(1, multiplier):
            t = torch.tensor([k / multiplier]).view(1, 1, 1, 1)
            out.append(m2m_forward(sd, x[i:i + 1], x[i + 1:i + 2], [t])[0])
    out.append(x[-1:])
    return out


def m2m_vfi(sd, frames, multiplier=2, states=None):
    """Node-level oracle: generic_frame_loop (vfi_utils.py:339-389).  int multiplier: one loop over the clip;
    list multiplier: one 2-frame loop per pair (m == 0 drops the pair, the last frame is kept only by the last pair,
    and the skip list sees local pair index 0)."""
    x = frames[..., :3].permute(0, 3, 1, 2).float()
    with torch.inference_mode():
        if type(multiplier) == int:
            out = _m2m_loop(sd, x, multiplier, states)
        else:
            ms = list(map(int, multiplier))
            ms += [2] * (len(x) - len(ms) - 1)
            out = []
            for i in range(len(x) - 1):
                if ms[i] == 0:
                    continue
                part = _m2m_loop(sd, x[i:i + 2], ms[i], states)
                out.extend(part if i == len(x) - 2 else part[:-1])
    return torch.cat(out, 0).permute(0, 2, 3, 1).contiguous()
