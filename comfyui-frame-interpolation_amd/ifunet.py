"""IFUnet VFI node (second half) — host-side mirror of vfi_models/ifunet/__init__.py over the HIP library.

IFUNetModel.forward (vfi_models/ifunet/IFUNet_arch.py:753-766) = flownet (IFUNet: a CBAM U-Net "FeatureNet" feeding three
convex-up-sampling IFBlocks, optionally evaluated a second time with the frames swapped, :654-743) -> fusionnet (RRDBNet blend
mask at quarter resolution, :269-328) -> refinenet (ResynNet: both frames re-aligned to the merged frame by three BatchNorm
FlowBlocks and blended with it, :117-193).  Every convolution runs on the fp32-MFMA layer objects (BatchNorm folded into the
weights at load time); CBAM's gates, the 4-channel convex up-sampling and the blends are the kernels of csrc/ifunet_ops.hip;
every ``torch.cat`` is a channel window of a pre-allocated NHWC tensor.

STATUS: oracle bit-exact vs the reference (oracle/VALIDATION_IFUNET.log); on the MI355X every pixel of a 1080p frame is within
1e-3 of the oracle and the node matches the reference's goldens (tests/test_gpu_ifunet.py, part of the default -m gpu suite since
round 2); the orchestration is also checked through the CPU test double (tests/test_ifunet_engine_cpu.py).
"""
import typing

import torch

from .ifunet_spec import CKPT_NAMES, ifunet_shapes
from .opsengine import OpsEngine, _p
from .schedule import InterpolationStateList, generic_output_plan

MODEL_TYPE = "ifunet"
LEVELS = (16, 8, 4)            # IFBlock up-sampling factors = FeatureNet output strides (:658-661)
WIDTHS = (256, 128, 64)


class IFUNetEngine(OpsEngine):
    def __init__(self, state_dict, device=None, _test_backend=None):
        super().__init__(device, _test_backend, pooled=True)      # scratch from a pool, recycled stage by stage (see forward)
        want = ifunet_shapes()
        missing = [k for k in want if k not in state_dict]
        if missing:
            raise KeyError(f"IFUNet checkpoint: missing keys {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for k, shp in want.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"IFUNet checkpoint: {k} has shape {tuple(state_dict[k].shape)}, expected {tuple(shp)}")
        self.scale, self.ensemble, self._pair = 1.0, True, None
        self._build(state_dict)

    # ---- layer objects ------------------------------------------------------------------------------------------------
    def _build(self, sd):
        mk = self._layer
        dev = lambda t: t.detach().to(self.device, torch.float32).contiguous()   # noqa: E731

        def cp(p, stride=1, **k):          # conv(): Conv2d + PReLU
            return mk(sd[p + ".0.weight"], sd[p + ".0.bias"], sd[p + ".1.weight"], stride=stride, **k)

        def cbp(p, stride=1, **k):         # conv_bn(): Conv2d(no bias) + BatchNorm2d(eval) + PReLU, folded
            s = sd[p + ".1.weight"] / torch.sqrt(sd[p + ".1.running_var"] + 1e-5)
            return mk(sd[p + ".0.weight"], None, sd[p + ".2.weight"], stride=stride, scale_out=s,
                      shift_out=sd[p + ".1.bias"] - sd[p + ".1.running_mean"] * s, **k)

        def cbam(p):
            if p + ".ChannelGate.mlp.1.weight" not in sd:
                return None
            q = p + ".SpatialGate.spatial."
            a = float(sd[q + "bn.weight"] / torch.sqrt(sd[q + "bn.running_var"] + 1e-5))
            return dict(w1=dev(sd[p + ".ChannelGate.mlp.1.weight"]), b1=dev(sd[p + ".ChannelGate.mlp.1.bias"]),
                        w2=dev(sd[p + ".ChannelGate.mlp.3.weight"]), b2=dev(sd[p + ".ChannelGate.mlp.3.bias"]),
                        w7=dev(sd[q + "conv.weight"][0].permute(1, 2, 0)), a=a, b=float(sd[q + "bn.bias"] - sd[q + "bn.running_mean"] * a))

        f = "flownet.fmap"
        self.f_conv0 = cp(f + ".conv0")                                        # 1x1, 7 -> 17
        self.f_down = [(cp(f"{f}.conv{i}.conv1", 2, cin_phys=24 if i == 1 else None), cp(f"{f}.conv{i}.conv2"), cbam(f"{f}.conv{i}.cbam"))
                       for i in range(1, 6)]
        self.f_up = {}
        for i in (5, 4, 3):
            q = f"{f}.deconv{i}"
            self.f_up[i] = (mk(sd[q + ".deconv.0.weight"], sd[q + ".deconv.0.bias"], sd[q + ".deconv.1.weight"], kind=1, stride=2),
                            cp(q + ".conv1"), cp(q + ".conv2"), cbam(q + ".cbam"))
        self.blocks = []
        for b, lvl in enumerate(LEVELS):
            q = f"flownet.block{b}"
            self.blocks.append(([cp(f"{q}.convblock.{i}") for i in range(6)], mk(sd[q + ".flowconv.weight"], sd[q + ".flowconv.bias"]),
                                mk(sd[f"{q}.maskconvx{lvl}.weight"], sd[f"{q}.maskconvx{lvl}.bias"])))
        q = "fusionnet"
        fifth = torch.full((64,), 0.2)
        self.r_first = mk(sd[q + ".conv_first.weight"], sd[q + ".conv_first.bias"])
        self.r_body = [[[mk(sd[f"{q}.body.{b}.rdb{r}.conv{k}.weight"], sd[f"{q}.body.{b}.rdb{r}.conv{k}.bias"], scale_out=fifth if k == 5 else None)
                         for k in range(1, 6)] for r in (1, 2, 3)] for b in range(6)]
        self.r_tail = {n: mk(sd[f"{q}.{n}.weight"], sd[f"{q}.{n}.bias"]) for n in ("conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last")}
        q = "refinenet"
        self.s_blocks = []
        for b in range(3):
            p = f"{q}.block{b}"
            self.s_blocks.append(([cbp(f"{p}.conv0.{i}", 2, cin_phys=16 if i == 0 else None) for i in range(3)],
                                  [cbp(f"{p}.convblock.{i}") for i in range(6)],
                                  mk(sd[p + ".lastconv.weight"], sd[p + ".lastconv.bias"], kind=1, stride=2)))
        self.s_ctx = [(cp(f"{q}.context{k}.0", 2, cin_phys=8), cp(f"{q}.context{k}.1", 2)) for k in (0, 1)]
        self.s_dec = (mk(sd[q + ".decode.0.weight"], sd[q + ".decode.0.bias"], kind=1, stride=2),
                      mk(sd[q + ".decode.1.weight"], sd[q + ".decode.1.bias"], kind=1, stride=2))

    # ---- composite ops --------------------------------------------------------------------------------------------------
    def _cbam(self, P, x, tag):
        """CBAM.forward (:499-503) -> new tensor"""
        n, h, w, c = x.shape
        stats, scale = self._t("cb_stats_" + tag, n, c, 2), self._t("cb_scale_" + tag, n, c)
        ws = self._t("cb_ws_" + tag, n * 512 * c * 3)      # (per pass: two ensemble passes of block 0 run side by side)
        self._c("vfi_channel_pool", _p(x), c, c, n, h * w, _p(stats), ws.data_ptr(), ws.numel() * 4)
        self._c("vfi_cbam_gate", _p(stats), _p(P["w1"]), _p(P["b1"]), _p(P["w2"]), _p(P["b2"]), c, P["w1"].shape[0], n, _p(scale))
        xs, comp = self._t("cb_xs_" + tag, n, h, w, c), self._t("cb_comp_" + tag, n * h * w, 2)
        self._c("vfi_cbam_scale_compress", _p(x), c, _p(scale), c, n, h * w, _p(xs), c, _p(comp))
        self._c("vfi_cbam_spatial", _p(xs), c, _p(comp), _p(P["w7"]), P["a"], P["b"], c, n, h, w)
        return xs

    def _feature_net(self, x, level, tag):
        """FeatureNet.forward (:582-597); x [1,h,w,24] holding 17 channels, or [1,h,w,8] holding 7 (block 0)"""
        n, h, w, cs = x.shape
        if cs == 8:
            x17 = self._t("fn_x17_" + tag, n, h, w, 24)
            self._conv(self.f_conv0, x, 0, x17, 0)
            x = x17
        skips = []
        for i, (c1, c2, att) in enumerate(self.f_down):
            h, w = h // 2, w // 2
            a, b = self._t(f"fn_a{i}_{tag}", n, h, w, c1["cout"]), self._t(f"fn_b{i}_{tag}", n, h, w, c1["cout"])
            self._conv(c1, x, 0, a, 0)
            self._conv(c2, a, 0, b, 0)
            x = self._cbam(att, b, f"d{i}_{tag}") if att is not None else b
            skips.append(x)
        y = skips[4]
        for i, skip in ((5, skips[3]), (4, skips[2]), (3, skips[1]))[:level + 1]:
            de, c1, c2, att = self.f_up[i]
            n, h, w, c = skip.shape
            cat = self._t(f"fn_cat{i}_{tag}", n, h, w, 2 * c)
            self._conv(de, y, 0, cat, 0)                       # deconv + PReLU -> first half of the concat
            self._ax(skip, 0, None, 0, cat, c, c)
            a, b = self._t(f"fn_ua{i}_{tag}", n, h, w, c1["cout"]), self._t(f"fn_ub{i}_{tag}", n, h, w, c2["cout"])
            self._conv(c1, cat, 0, a, 0)
            self._conv(c2, a, 0, b, 0)
            y = self._cbam(att, b, f"u{i}_{tag}") if att is not None else b
        return y

    def _if_block(self, i, fmap, out, tag):
        """IFBlock.forward (:640-651) at scale 1/s -> flow delta [1,Hp,Wp,4] written to ``out``"""
        convs, flowconv, maskconv = self.blocks[i]
        n, h, w, c = fmap.shape
        p, q = self._t(f"ib_p{i}_{tag}", n, h, w, c), self._t(f"ib_q{i}_{tag}", n, h, w, c)
        cur = fmap
        for k, L in enumerate(convs):
            nxt = p if k % 2 == 0 else q
            self._conv(L, cur, 0, nxt, 0)
            cur = nxt
        x = p if cur is q else q
        self._ax(cur, 0, fmap, 0, x, 0, c)                     # convblock(x) + x
        lvl = LEVELS[i]
        fl, mk_ = self._t(f"ib_f{i}_{tag}", n, h, w, 8), self._t(f"ib_m{i}_{tag}", n, h, w, 9 * lvl * lvl)
        self._conv(flowconv, x, 0, fl, 0)
        self._conv(maskconv, x, 0, mk_, 0)
        up = self._t(f"ib_up_{tag}", n, h * lvl, w * lvl, 4)
        self._c("vfi_convex_upsample_c", _p(mk_), mk_.shape[-1], _p(fl), 8, _p(up), 4, n, h, w, lvl, 4)
        inv = out.shape[1] / up.shape[1]                       # F.interpolate(flow_up, scale_factor=1/s) * (1/s)
        self._resize(up, 0, out, 0, 4, inv)

    def _scaled_sizes(self, Hp, Wp, s):
        hs, ws = Hp * s, Wp * s
        if abs(hs - round(hs)) > 1e-9 or abs(ws - round(ws)) > 1e-9 or round(hs) % 32 or round(ws) % 32 or round(hs) < 32 or round(ws) < 32:
            raise NotImplementedError(f"IFUNet (HIP): scale {s} on a {Hp}x{Wp} padded frame does not give a multiple-of-32 working size; "
                                      f"only such scales are on the device path (the reference needs /32 sizes as well)")
        hs, ws = int(round(hs)), int(round(ws))
        if float(Hp) / hs != 1.0 / s and abs(Hp / hs - 1.0 / s) > 1e-12:
            raise NotImplementedError(f"IFUNet (HIP): scale {s} is not exactly representable as a size ratio")
        return hs, ws

    # ---- IFUNetModel.forward -------------------------------------------------------------------------------------------
    fork_stages = True      # independent stages of a call on two streams (tests A/B it; frames are bit-identical)

    def lone_pair(self, on):
        """lanes.tell_lone_pair: fork only when this is the only pair in flight"""
        self.fork_stages = bool(on)

    def forward(self, frame0, frame1, t, out, scale=None, ensemble=None):
        s = float(self.scale if scale is None else scale)
        ens = bool(self.ensemble if ensemble is None else ensemble)
        t = float(t)
        assert frame1.shape == frame0.shape and frame0.shape[2] >= 3 and frame0.is_contiguous() and frame1.is_contiguous()
        # one HIP graph per (frame shape, timestep, scale, ensemble): the ~2000 launches of a call replay without the interpreter
        self._replayable(("forward",) + tuple(frame0.shape) + (t, s, ens, self.fork_stages), (frame0, frame1), (out,),
                         lambda f0, f1, o: self._forward(f0, f1, t, o, s, ens))
        return out

    def _forward(self, frame0, frame1, t, out, s, ens):
        H, W, Cc = frame0.shape
        Hp, Wp = ((H - 1) // 64 + 1) * 64, ((W - 1) // 64 + 1) * 64
        hs, ws = self._scaled_sizes(Hp, Wp, s)
        px = Hp * Wp
        with self._scope():      # everything of a call is scratch: the pool is empty again when it returns
            # inputs of the three levels: x17 = (img0, img1, t, w0, w1, flow), x17e the swapped "ensemble" variant
            x17, x17e = self._t("x17", 1, Hp, Wp, 24), self._t("x17e", 1, Hp, Wp, 24)
            x7, x7e = self._t("x7", 1, Hp, Wp, 8), self._t("x7e", 1, Hp, Wp, 8)
            for dst, order, tt in ((x17, (frame0, frame1), t), (x17e, (frame1, frame0), 1 - t), (x7, (frame0, frame1), t), (x7e, (frame1, frame0), 1 - t)):
                for k, fr in enumerate(order):
                    self._c("vfi_pad_rgb", fr.data_ptr(), Cc, H, W, _p(dst, 3 * k), dst.shape[-1], Hp, Wp)
                self._c("vfi_fill_channels", _p(dst, 6), dst.shape[-1], 1, px, tt)
            flow, delta, flow2 = self._t("flow", 1, Hp, Wp, 4), self._t("delta", 1, Hp, Wp, 4), self._t("flow2", 1, Hp, Wp, 4)

            def estimate(i, xin, nimg, with_flow, tag, dst=None):
                if s != 1.0:
                    xs = self._t(f"xs{nimg}_{tag}", 1, hs, ws, xin.shape[-1])
                    self._resize(xin, 0, xs, 0, nimg)
                    if with_flow:
                        self._resize(flow, 0, xs, 13, 4, s)            # F.interpolate(flow, scale) * scale
                else:
                    xs = xin
                    if with_flow:
                        self._ax(flow, 0, None, 0, xs, 13, 4)
                self._if_block(i, self._feature_net(xs, i, tag), delta if dst is None else dst, tag)

            def stage(fn, *a):      # the temporaries of one stage (feature net + IFBlock, RRDBNet, a ResynNet pass) die with it
                with self._scope():
                    return fn(*a)

            for i in range(3):
                if i == 0 and ens and self.fork_stages and self._fork()[1] is not None:
                    # r6: block 0 takes no flow, so its two ensemble passes are independent: the swapped one runs on the side stream
                    # (one scope holds the temporaries of both until the join)
                    with self._scope():
                        cur, side = self._fork()
                        delta_b = self._t("delta_b", 1, Hp, Wp, 4)
                        with torch.cuda.stream(side):
                            estimate(0, x7e, 7, False, "b", delta_b)
                        estimate(0, x7, 7, False, "a")
                        self._join(cur, side)
                        self._ax(delta, 0, None, 0, flow, 0, 4)
                        self._ax(flow, 0, delta_b, 0, flow, 0, 4, 0.5, 0.5)        # (flow + flow2) / 2
                elif i == 0:
                    stage(estimate, 0, x7, 7, False, "a")
                    self._ax(delta, 0, None, 0, flow, 0, 4)
                    if ens:
                        stage(estimate, 0, x7e, 7, False, "b")
                        self._ax(flow, 0, delta, 0, flow, 0, 4, 0.5, 0.5)          # (flow + flow2) / 2
                else:
                    stage(estimate, i, x17, 13, True, "a")
                    self._ax(flow, 0, delta, 0, flow, 0, 4)
                    if ens:
                        stage(estimate, i, x17e, 13, True, "b")
                        self._ax(flow, 0, delta, 0, flow2, 0, 4)                   # flow2 = flow + flow_d
                        self._ax(flow, 0, flow2, 0, flow, 0, 4, 0.5, 0.5)
                for k in (0, 1):   # warped_img0 = warp(img0, flow[:, :2]), warped_img1 = warp(img1, flow[:, 2:4]); same slots in both variants
                    self._c("vfi_warp_rife", _p(x17, 3 * k), 24, _p(flow, 2 * k), 4, _p(x17, 7 + 3 * k), 24, 1, Hp, Wp, 3)
                self._ax(x17, 7, None, 0, x17e, 7, 6)
            mask = self._t("rr_mask", 1, Hp, Wp, 8)              # a stage's outputs are requested here first, so that they outlive its scope
            stage(self._rrdbnet, x17, flow)
            deg = self._t("deg", 1, Hp, Wp, 4)
            self._c("vfi_lerp_mask", _p(x17, 7), 24, _p(x17, 10), 24, _p(mask), mask.shape[-1], _p(deg), 4, 3, px)
            imgs = [self._t(f"rs_img_i{k}", 1, Hp, Wp, 4) for k in (0, 1)]
            masks = [self._t(f"rs_mask_i{k}", 1, Hp, Wp, 1) for k in (0, 1)]
            # r6: the two ResynNet passes (image 0, image 1) are independent: the second runs on the engine's side stream beside the first.
            # ONE scope holds the temporaries of both until the join (a scope that ended earlier would hand its blocks to the other pass).
            with self._scope():
                cur, side = self._fork() if self.fork_stages else (None, None)
                if side is not None:
                    with torch.cuda.stream(side):
                        self._resyn(x17, 3, deg, "i1")
                    self._resyn(x17, 0, deg, "i0")
                    self._join(cur, side)
                else:
                    self._resyn(x17, 0, deg, "i0")
                    self._resyn(x17, 3, deg, "i1")
            self._c("vfi_ifunet_blend", _p(imgs[0]), _p(imgs[1]), _p(deg), 4, _p(masks[0]), _p(masks[1]), 1, out.data_ptr(), Hp, Wp, H, W)
            return out

    def _rrdbnet(self, x17, flow):
        """RRDBNet.forward (:306-328) -> sigmoid mask [1,Hp,Wp,8] (channel 0)"""
        _, Hp, Wp, _ = x17.shape
        h, w = Hp // 4, Wp // 4
        r16 = self._t("rr_in", 1, h, w, 16)
        self._resize(x17, 0, r16, 0, 6)                    # img0, img1
        self._resize(x17, 7, r16, 6, 6)                    # warped_img0, warped_img1
        self._resize(flow, 0, r16, 12, 4, 0.25)
        feat = self._t("rr_feat", 1, h, w, 64)
        self._conv(self.r_first, r16, 0, feat, 0)
        # three rotating [x | x_1 .. x_4] buffers: a dense block's output goes to the buffer that holds neither its input nor the RRDB's
        # input, so the RRDB's input survives to its closing "out * 0.2 + x" without a copy (r6: 6 copies of 67 MB per frame gone)
        d = [self._t("rr_d0", 1, h, w, 192), self._t("rr_d1", 1, h, w, 192), self._t("rr_d2", 1, h, w, 192)]
        self._ax(feat, 0, None, 0, d[0], 0, 64)
        cur = 0
        for blk in self.r_body:
            start = cur
            for rdb in blk:                                # x_k = lrelu(conv_k(cat(x, x_1..x_{k-1}))) at offset 64 + 32 (k-1)
                for k in range(4):
                    self._conv(rdb[k], d[cur], 0, d[cur], 64 + 32 * k, act=1, slope=0.2)
                nxt = next(j for j in range(3) if j != cur and j != start)
                self._conv(rdb[4], d[cur], 0, d[nxt], 0, res=d[cur])          # conv5 * 0.2 + x  (0.2 folded into the layer)
                cur = nxt
            self._ax(d[cur], 0, d[start], 0, d[cur], 0, 64, 0.2, 1.0)        # out * 0.2 + x
        body = self._t("rr_body", 1, h, w, 64)
        self._conv(self.r_tail["conv_body"], d[cur], 0, body, 0, res=feat)     # feat + conv_body(body)
        x = body
        for name in ("conv_up1", "conv_up2"):
            n, hh, ww, _ = x.shape
            up = self._t("rr_up_" + name, 1, 2 * hh, 2 * ww, 64)
            self._c("vfi_upsample_nearest", _p(x), 64, _p(up), 64, 1, hh, ww, 2 * hh, 2 * ww, 64)
            y = self._t("rr_y_" + name, 1, 2 * hh, 2 * ww, 64)
            self._conv(self.r_tail[name], up, 0, y, 0, act=1, slope=0.2)
            x = y
        hr = self._t("rr_hr", 1, Hp, Wp, 64)
        self._conv(self.r_tail["conv_hr"], x, 0, hr, 0, act=1, slope=0.2)
        mask = self._t("rr_mask", 1, Hp, Wp, 8)
        self._conv(self.r_tail["conv_last"], hr, 0, mask, 0, act=4)
        return mask

    def _resyn(self, x17, ioff, deg, tag):
        """ResynNet.calflow (:139-161) for the image at channels ioff..ioff+2 of x17 -> (refined image [1,Hp,Wp,4], mask [1,Hp,Wp,1])"""
        _, Hp, Wp, _ = x17.shape
        px = Hp * Wp
        T = lambda name, *shape: self._t(f"{name}_{tag}", *shape)      # every temporary carries the pass's tag: the two passes run side by side (r6)
        y = T("rs_y", 1, Hp, Wp, 16)          # img 3 | deg 3 | warped 3 | mask 1 | (flow 2 after the resize) | pad
        self._ax(x17, ioff, None, 0, y, 0, 3)
        self._ax(deg, 0, None, 0, y, 3, 3)
        flow, mask = T("rs_flow", 1, Hp, Wp, 2), T("rs_mask", 1, Hp, Wp, 1)
        df, dm = T("rs_df", 1, Hp, Wp, 2), T("rs_dm", 1, Hp, Wp, 1)
        for b, sc in enumerate((4, 2, 1)):
            conv0, convblock, last = self.s_blocks[b]
            h, w = Hp // sc, Wp // sc
            xs = T(f"rs_xs{b}", 1, h, w, 16)
            self._resize(y, 0, xs, 0, 6 if b == 0 else 10)
            if b > 0:
                self._resize(flow, 0, xs, 10, 2, 1.0 / sc)
            cur = xs
            for i, L in enumerate(conv0):
                h, w = h // 2, w // 2
                nxt = T(f"rs_c{b}_{i}", 1, h, w, L["cout"])
                self._conv(L, cur, 0, nxt, 0)
                cur = nxt
            feat = cur
            p, q = T(f"rs_p{b}", 1, h, w, 256), T(f"rs_q{b}", 1, h, w, 256)
            for i, L in enumerate(convblock):
                nxt = p if i % 2 == 0 else q
                self._conv(L, cur, 0, nxt, 0)
                cur = nxt
            s = p if cur is q else q
            self._ax(cur, 0, feat, 0, s, 0, 256)                                # convblock(feat) + feat
            tmp = T(f"rs_t{b}", 1, 2 * h, 2 * w, 8)
            self._conv(last, s, 0, tmp, 0)
            self._resize(tmp, 0, df, 0, 2, float(sc * 4))                       # tmp[:, :2] * scale * 4 after interpolate(scale * 4)
            self._resize(tmp, 2, dm, 0, 1)
            if b == 0:
                self._ax(df, 0, None, 0, flow, 0, 2)
                self._ax(dm, 0, None, 0, mask, 0, 1)
            else:
                self._ax(flow, 0, df, 0, flow, 0, 2)
                self._ax(mask, 0, dm, 0, mask, 0, 1)
            self._c("vfi_warp_rife", _p(y, 0), 16, _p(flow), 2, _p(y, 6), 16, 1, Hp, Wp, 3)     # warped_img0 = warp(img0, flow)
            self._ax(mask, 0, None, 0, y, 9, 1)
        h, w = Hp // 4, Wp // 4
        fdown = T("rs_fdown", 1, h, w, 2)
        self._resize(flow, 0, fdown, 0, 2, 0.25)
        cat = T("rs_cat", 1, h, w, 64)
        for k, (c0, c1) in enumerate(self.s_ctx):           # context0(img0) warped by the down-scaled flow | context1(warped_img0)
            src = T(f"rs_src{k}", 1, Hp, Wp, 8)
            self._ax(y, 0 if k == 0 else 6, None, 0, src, 0, 3)
            a, bq = T(f"rs_ca{k}", 1, Hp // 2, Wp // 2, 16), T(f"rs_cb{k}", 1, h, w, 32)
            self._conv(c0, src, 0, a, 0)
            self._conv(c1, a, 0, bq, 0)
            if k == 0:
                self._c("vfi_warp_rife", _p(bq), 32, _p(fdown), 2, _p(cat, 0), 64, 1, h, w, 32)
            else:
                self._ax(bq, 0, None, 0, cat, 32, 32)
        d1, d2 = T("rs_d1", 1, Hp // 2, Wp // 2, 32), T("rs_d2", 1, Hp, Wp, 8)
        self._conv(self.s_dec[0], cat, 0, d1, 0)
        self._conv(self.s_dec[1], d1, 0, d2, 0)
        self._c("vfi_tanh_scale", _p(d2), 8, 3, px, 1.0)
        img = T("rs_img", 1, Hp, Wp, 4)
        self._c("vfi_add_clamp01", _p(y, 6), 16, _p(d2), 8, _p(img), 4, 3, px)
        return img, mask

    # (pair, timestep) interface of m2m.run_plan
    def prepare(self, frame0, frame1):
        self._pair = (frame0, frame1)

    def render(self, t, out):
        return self.forward(self._pair[0], self._pair[1], t, out)

    def release_workspace(self):
        super().release_workspace()
        self._pair = None


class IFUnet_VFI:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (CKPT_NAMES,),
                "frames": ("IMAGE",),
                "clear_cache_after_n_frames": ("INT", {"default": 10, "min": 1, "max": 1000}),
                "multiplier": ("INT", {"default": 2, "min": 2, "max": 1000}),
                "scale_factor": ("FLOAT", {"default": 1.0, "min": 0.1, "max": 100, "step": 0.1}),
                "ensemble": ("BOOLEAN", {"default": True}),
            },
            "optional": {"optional_interpolation_states": ("INTERPOLATION_STATES",)},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "vfi"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def vfi(self, ckpt_name: typing.AnyStr, frames: torch.Tensor, clear_cache_after_n_frames: typing.SupportsInt = 1,
            multiplier: typing.SupportsInt = 2, scale_factor: typing.SupportsFloat = 1.0, ensemble: bool = True,
            optional_interpolation_states: InterpolationStateList = None, **kwargs):
        from .ckpt import load_file_from_github_release
        from .m2m import run_plan

        assert len(frames) >= 2, f"VFI model IFUNet requires at least 2 frames to work with, only found {frames.shape[0]}."
        model_path = load_file_from_github_release(MODEL_TYPE, ckpt_name)
        from .ckpt import begin_call, cached_engine, end_call
        from .lanes import configure as configure_lanes
        from .lanes import lane_set

        def build():
            sd = torch.load(model_path, map_location="cpu", weights_only=False)
            return lane_set("ifunet", lambda: IFUNetEngine(sd))
        # (the reference rebuilds the model on every call; here the packed weights stay between calls — the constructor, Winograd weight
        # transforms of ~250 layers, is 0.4 s per lane; see ckpt.cached_engine)
        engine, cached = cached_engine(MODEL_TYPE, model_path, build)
        sc, ens = float(scale_factor), bool(ensemble)
        configure_lanes(engine, lambda e: (setattr(e, "scale", sc), setattr(e, "ensemble", ens)))
        try:
            begin_call(engine, frames.shape[1:3])
            plan, tasks = generic_output_plan(len(frames), multiplier, optional_interpolation_states)
            return (run_plan(engine, frames, plan, tasks, name="IFUnet VFI"),)
        finally:
            torch.cuda.synchronize(engine.device)
            end_call(engine, cached)      # (the workspace and the captured graphs stay for the next call of this frame shape: ckpt.KEEP_WORKSPACE_BYTES)
