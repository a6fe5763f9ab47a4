"""GMFSS Fortuna VFI node, "union" and base models — host-side mirror of vfi_models/gmfss_fortuna/__init__.py over
the HIP library.  FIRST-CORRECT device path: every convolution / linear layer runs on the fp32-MFMA layer objects
(vfi_conv_*), everything else on the one-thread-per-element kernels of csrc/gmfss_ops.hip (their bodies are checked on the
host by the CPU test suite, tests/test_gmfss_bodies_cpu.py; the orchestration below by tests/test_gmfss_engine_cpu.py
through a test double of the C ABI).  The attention / matching products are plain VALU loops for now.

Model.reuse (GMFSS_Fortuna_union_arch.py:1726-1782) = ``prepare(frame0, frame1)``: FeatureNet on both frames, GMFlow in both
directions on the half-resolution pair, MetricNet.  The backbone and the first (global-matching) scale of GMFlow are
symmetric in the two frames, so they are evaluated once for both directions; the second scale runs as a batch of two.
Model.inference (:1784-1857) = ``render(t)``: eight soft splats, IFNet 4.6 on the half-resolution pair, GridNet, clamp.
Every ``torch.cat`` is a channel window of a pre-allocated NHWC tensor.
"""
import ctypes as C
import math
import typing

import torch

from .gmfss_spec import gmfss_shapes
from .schedule import InterpolationStateList, generic_output_plan

MODEL_TYPE = "gmfss_fortuna"
CKPTS_PATH_CONFIG = {   # gmfss_fortuna/__init__.py:11-26
    "GMFSS_fortuna_union": {
        "ifnet": ("rife", "rife46.pth"),
        "flownet": (MODEL_TYPE, "GMFSS_fortuna_flownet.pkl"),
        "metricnet": (MODEL_TYPE, "GMFSS_fortuna_union_metric.pkl"),
        "feat_ext": (MODEL_TYPE, "GMFSS_fortuna_union_feat.pkl"),
        "fusionnet": (MODEL_TYPE, "GMFSS_fortuna_union_fusionnet.pkl"),
    },
    "GMFSS_fortuna": {   # base model: no IFNet, GridNet head on (img0, I1t, I2t, img1)
        "flownet": (MODEL_TYPE, "GMFSS_fortuna_flownet.pkl"),
        "metricnet": (MODEL_TYPE, "GMFSS_fortuna_metric.pkl"),
        "feat_ext": (MODEL_TYPE, "GMFSS_fortuna_feat.pkl"),
        "fusionnet": (MODEL_TYPE, "GMFSS_fortuna_fusionnet.pkl"),
    },
}
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


from .opsengine import OpsEngine, _cs, _p


# ---- host-side constant tables (uploaded once per shape) ------------------------------------------------------------
def sine_position(c, h, w, temperature=10000.0):
    """PositionEmbeddingSine (:1015-1056) for a [h, w] window, num_pos_feats = c/2, normalize=True -> [h, w, c]"""
    npf = c // 2
    ones = torch.ones((1, h, w))
    y = ones.cumsum(1, dtype=torch.float32)
    x = ones.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = x / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    d = torch.arange(npf, dtype=torch.float32)
    d = temperature ** (2 * (d // 2) / npf)
    px, py = x[:, :, :, None] / d, y[:, :, :, None] / d
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3)[0].contiguous()


def shift_mask(h, w, k):
    """generate_shift_window_attn_mask (:326-364) -> [k*k, lw, lw] with lw = (h/k)*(w/k)"""
    wh, ww = h // k, w // k

    def region(n, win):
        r = torch.zeros(n)
        r[n - win:n - win // 2] = 1
        r[n - win // 2:] = 2
        return r

    label = region(h, wh)[:, None] * 3 + region(w, ww)[None, :]
    win = label.view(k, wh, k, ww).permute(0, 2, 1, 3).reshape(k * k, wh * ww)
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).contiguous()


def shift_labels(h, w, k):
    """The same mask as region labels per token in window order, int32 [k*k, lw]: two tokens of a (shifted) window may attend
    to each other iff their labels are equal — what vfi_attention takes instead of the [k*k, lw, lw] float mask."""
    wh, ww = h // k, w // k

    def region(n, win):
        r = torch.zeros(n, dtype=torch.int32)
        r[n - win:n - win // 2] = 1
        r[n - win // 2:] = 2
        return r

    label = region(h, wh)[:, None] * 3 + region(w, ww)[None, :]
    return label.view(k, wh, k, ww).permute(0, 2, 1, 3).reshape(k * k, wh * ww).contiguous()


class GMFSSEngine(OpsEngine):
    def __init__(self, state_dicts, device=None, _test_backend=None):
        super().__init__(device, _test_backend, pooled=True)      # 1080p: 9.9 GiB of named scratch tensors without the pool
        self.union = "ifnet" in state_dicts      # the union model carries rife46.pth; the base model has no IFNet
        shapes = gmfss_shapes("union" if self.union else "base")
        for part in shapes:
            sd = state_dicts[part]
            missing = [k for k in shapes[part] if k not in sd]
            if missing:
                raise KeyError(f"GMFSS {part} checkpoint: missing keys {missing[:4]}{'...' if len(missing) > 4 else ''}")
            for k, shp in shapes[part].items():
                if tuple(sd[k].shape) != tuple(shp):
                    raise ValueError(f"GMFSS {part} checkpoint: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shp)}")
        self._build(state_dicts)
        self.prepared = None

    def release_workspace(self):
        super().release_workspace()
        self.prepared = None

    # ---- layer objects ----------------------------------------------------------------------------------------------
    def _build(self, sds):
        mk = self._layer
        # FeatureNet (:1470-1500): PReLU -> conv s2 -> PReLU -> conv; the 2nd PReLU is the first conv's epilogue
        fe = sds["feat_ext"]
        self.fe = [(float(fe[f"block{k}.0.weight"]), mk(fe[f"block{k}.1.weight"], fe[f"block{k}.1.bias"], fe[f"block{k}.2.weight"], stride=2,
                                                       cin_phys=8 if k == 1 else None),
                    mk(fe[f"block{k}.3.weight"], fe[f"block{k}.3.bias"])) for k in (1, 2, 3)]
        # GMFlow backbone (:218-312): convs without bias, InstanceNorm between them
        fl = sds["flownet"]
        dev = lambda t: t.detach().to(self.device, torch.float32).contiguous()   # noqa: E731
        self.bb_head = (dev(fl["backbone.conv1.weight"].permute(2, 3, 1, 0)), dev(torch.zeros(64)), dev(torch.ones(64)))
        self.bb = {}
        for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 1)):
            for blk in (0, 1):
                p = f"backbone.{name}.{blk}."
                s = stride if blk == 0 else 1
                ds = mk(fl[p + "downsample.0.weight"], fl[p + "downsample.0.bias"]) if p + "downsample.0.weight" in fl else None
                self.bb[(name, blk)] = (mk(fl[p + "conv1.weight"], stride=s), mk(fl[p + "conv2.weight"]), ds, s)
        self.bb_conv2 = mk(fl["backbone.conv2.weight"], fl["backbone.conv2.bias"])
        self.trident = (mk(fl["backbone.trident_conv.weight"]), mk(fl["backbone.trident_conv.weight"], stride=2))
        # transformer (:439-686): Linear layers as 1x1 convs over [B,h,w,C] token maps
        self.tf = []
        for i in range(6):
            blk = {}
            for part in ("self_attn", "cross_attn_ffn"):
                p = f"transformer.layers.{i}.{part}."
                d = {n: mk(fl[p + n + ".weight"]) for n in ("q_proj", "merge")}
                # r5: the projections that share an input are ONE layer with concatenated output channels (every output channel's sum over
                # K is what it was): self part q | k | v (128 -> 384), cross part k | v (128 -> 256) — these Linear layers are launch-latency
                # bound (1 GFLOP each at 1080p), so three launches cost three times one
                if part == "self_attn":
                    d["qkv"] = mk(torch.cat([fl[p + n + ".weight"] for n in ("q_proj", "k_proj", "v_proj")], 0))
                else:
                    d["kv"] = mk(torch.cat([fl[p + n + ".weight"] for n in ("k_proj", "v_proj")], 0))
                d["norm1"] = (dev(fl[p + "norm1.weight"]), dev(fl[p + "norm1.bias"]))
                if part == "cross_attn_ffn":
                    d["mlp0"], d["mlp2"] = mk(fl[p + "mlp.0.weight"]), mk(fl[p + "mlp.2.weight"])
                    d["norm2"] = (dev(fl[p + "norm2.weight"]), dev(fl[p + "norm2.bias"]))
                blk[part] = d
            self.tf.append(blk)
        self.prop_q = mk(fl["feature_flow_attn.q_proj.weight"], fl["feature_flow_attn.q_proj.bias"])
        self.prop_k = mk(fl["feature_flow_attn.k_proj.weight"], fl["feature_flow_attn.k_proj.bias"])
        # up-sampler (:1195-1199): input cat(flow 2, feature 128) held as [feature 128 | flow 2 | pad]
        self.up0 = mk(fl["upsampler.0.weight"], fl["upsampler.0.bias"], chan_map=[128, 129] + list(range(128)), cin_phys=136)
        self.up2 = mk(fl["upsampler.2.weight"], fl["upsampler.2.bias"])
        # MetricNet (:1420-1467)
        mn = sds["metricnet"]
        self.m_in = mk(mn["metric_in.weight"], mn["metric_in.bias"], cin_phys=16)
        self.m_net = [(float(mn[f"metric_net{k}.0.weight"]), mk(mn[f"metric_net{k}.1.weight"], mn[f"metric_net{k}.1.bias"])) for k in (1, 2, 3)]
        self.m_out = (float(mn["metric_out.0.weight"]), mk(mn["metric_out.1.weight"], mn["metric_out.1.bias"]))
        # GridNet (:1503-1688): Sequential(PReLU, conv | deconv, PReLU, conv) blocks
        gn = sds["fusionnet"]

        def pair(p, transposed=False, stride=1, cin_phys=None):
            first = mk(gn[p + "1.weight"], gn[p + "1.bias"], gn[p + "2.weight"], kind=1 if transposed else 0, stride=2 if transposed else stride,
                       cin_phys=cin_phys)
            return float(gn[p + "0.weight"]), first, mk(gn[p + "3.weight"], gn[p + "3.bias"])

        self.gn = {}
        self.gn["reshead0"] = pair("residual_model_head0." if self.union else "residual_model_head.", cin_phys=16)   # 9 / 12 channels
        for name in ("head1", "head2", "head3", "01", "04", "05", "11", "14", "15", "21", "24", "25"):
            self.gn["res" + name] = pair(f"residual_model_{name}.")
        for name in ("10", "20", "11", "21"):
            self.gn["down" + name] = pair(f"downsample_model_{name}.", stride=2)
        for name in ("04", "14", "05", "15"):
            self.gn["up" + name] = pair(f"upsample_model_{name}.", transposed=True)
        p = "residual_model_tail."
        self.tail = (mk(gn[p + "conv_before_upsample.0.weight"], gn[p + "conv_before_upsample.0.bias"], gn[p + "conv_before_upsample.1.weight"]),
                     mk(gn[p + "upsample.0.weight"], gn[p + "upsample.0.bias"]), mk(gn[p + "conv_last.weight"], gn[p + "conv_last.bias"]))
        # IFNet 4.6 (rife_arch.py:187-276,404-408): ResConv's beta folded into weights and bias
        rf = sds.get("ifnet", {})
        self.rife = []
        for b in range(4 if self.union else 0):
            p = f"block{b}."
            self.rife.append(dict(
                c=rf[p + "conv0.1.0.weight"].shape[0],
                c00=mk(rf[p + "conv0.0.0.weight"], rf[p + "conv0.0.0.bias"], stride=2, cin_phys=8 if b == 0 else 16),
                c01=mk(rf[p + "conv0.1.0.weight"], rf[p + "conv0.1.0.bias"], stride=2),
                res=[mk(rf[p + f"convblock.{i}.conv.weight"], rf[p + f"convblock.{i}.conv.bias"], scale_out=rf[p + f"convblock.{i}.beta"])
                     for i in range(8)],
                last=mk(rf[p + "lastconv.0.weight"], rf[p + "lastconv.0.bias"], kind=1, stride=2)))

    # ---- small composite ops ----------------------------------------------------------------------------------------
    def _pair(self, blk, src, soff, cin, dst, doff, tag, res=None):
        """Sequential(PReLU(a), conv | deconv, PReLU(b), conv)(src[..., soff:soff+cin]) (+ res) -> dst[..., doff:]"""
        a, first, second = blk
        n, h, w, _ = src.shape
        with self._scope():
            t = self._t("pair_in_" + tag, n, h, w, _cs(cin))
            self._prelu(src, soff, t, 0, cin, a)
            ho, wo = (2 * h, 2 * w) if first["kind"] == 1 else (h // first["stride"], w // first["stride"])
            m = self._t("pair_mid_" + tag, n, ho, wo, _cs(first["cout"]))
            self._conv(first, t, 0, m, 0)
            self._conv(second, m, 0, dst, doff, res=res)

    def _instnorm(self, x, c, relu1, add, relu2, out):
        n, h, w, cs = x.shape
        stats = self._t("in_stats", n, c, 2)
        ws = self.scratch.setdefault(("in_ws", n, c), torch.zeros(n * 512 * c * 2, dtype=torch.float64, device=self.device))
        self._c("vfi_instnorm_stats", _p(x), cs, c, n, h * w, _p(stats), ws.data_ptr(), ws.numel() * 8)
        self._c("vfi_instnorm_apply", _p(x), cs, _p(stats), c, n, h * w, int(relu1), _p(add) if add is not None else None,
                add.shape[-1] if add is not None else 0, int(relu2), _p(out), out.shape[-1])

    def _res_block(self, key, x, cin):
        """ResidualBlock_class.forward (:207-215)"""
        c1, c2, ds, stride = self.bb[key]
        n, h, w, _ = x.shape
        ho, wo, c = h // stride, w // stride, c1["cout"]
        tag = f"{key[0]}{key[1]}"
        y = self._t(f"bb_{tag}_y", n, ho, wo, c)
        with self._scope():
            a, b = (self._t(f"bb_{tag}_{k}", n, ho, wo, c) for k in "ab")
            self._conv(c1, x, 0, a, 0)
            self._instnorm(a, c, True, None, False, a)
            self._conv(c2, a, 0, b, 0)
            short = x
            if ds is not None:
                xs = x
                if stride == 2:   # 1x1 conv with stride 2 = sample the even pixels, then the 1x1 conv
                    xs = self._t(f"bb_{tag}_sub", n, ho, wo, x.shape[-1])
                    self._c("vfi_upsample_nearest", _p(x), x.shape[-1], _p(xs), xs.shape[-1], n, h, w, ho, wo, cin)
                short = self._t(f"bb_{tag}_s", n, ho, wo, c)
                self._conv(ds, xs, 0, short, 0)
                self._instnorm(short, c, False, None, False, short)
            self._instnorm(b, c, True, short, True, y)
        return y

    def _attention(self, q, qoff, k, koff, v, voff, out, h, w, splits, shifted):
        """single_head_split_window_attention (:367-436) on [B,h,w,C] token maps; q / k / v are channel windows (offset, 128 wide) of
        tensors that may hold several projections side by side"""
        B, c = q.shape[0], out.shape[-1]
        wh, ww = h // splits, w // splits
        sh, sw = (wh // 2, ww // 2) if shifted else (0, 0)
        # softmax(q k^T / sqrt(c) + mask) v as one flash-style MFMA kernel (csrc/attention.hip): no score matrix in HBM, and the
        # roll / window split / merge / roll back are the kernel's addressing — no partitioned copies either
        labels = self._const(("labels", h, w, splits), lambda: shift_labels(h, w, splits)) if shifted else None
        self._c("vfi_window_attention", _p(q, qoff), q.shape[-1], _p(k, koff), k.shape[-1], _p(v, voff), v.shape[-1], _p(out), out.shape[-1], B, h, w, splits,
                sh, sw, c, 1.0 / c ** 0.5, _p(labels) if shifted else None)

    def _pair_swap(self, a, o):
        """o[2d + s] = a[2d + 1 - s]: concat1 of FeatureTransformer.forward (:664-678)"""
        for i in range(a.shape[0]):
            self._ax(a[i ^ 1:(i ^ 1) + 1], 0, None, 0, o[i:i + 1], 0, a.shape[-1])

    def _transformer(self, a, splits):
        """FeatureTransformer.forward on a [2D, h, w, 128] (per direction: source, target), in place"""
        B, h, w, c = a.shape
        with self._scope():      # q / k / v / message / FFN buffers: 1.2 GiB at 1080p, dead when the transformer returns
            q, m = (self._t("tf_" + n, B, h, w, c) for n in ("q", "m"))
            qkv = self._t("tf_qkv", B, h, w, 3 * c)      # self part: q | k | v of one launch
            kvx = self._t("tf_kvx", B, h, w, 2 * c)      # cross part: k | v of the OTHER image of the pair
            cat = self._t("tf_cat", B, h, w, 2 * c)
            hid = self._t("tf_hid", B, h, w, 8 * c)
            for i, blk in enumerate(self.tf):
                shifted = i % 2 == 1
                # The cross part attends to the OTHER image's features as they were when the block started (concat1 is rebuilt only
                # after a whole block, :664-678): its key / value projections are taken here, before the self part updates `a`, with
                # the (source, target) swap as the batch index of the projection's input — no swapped copy of the features.
                for j in range(B):
                    self._conv(blk["cross_attn_ffn"]["kv"], a[j ^ 1:(j ^ 1) + 1], 0, kvx[j:j + 1], 0)
                for part in ("self_attn", "cross_attn_ffn"):
                    L = blk[part]
                    ffn = part == "cross_attn_ffn"
                    if ffn:
                        self._conv(L["q_proj"], a, 0, q, 0)
                        self._attention(q, 0, kvx, 0, kvx, c, m, h, w, splits, shifted)
                    else:
                        self._conv(L["qkv"], a, 0, qkv, 0)
                        self._attention(qkv, 0, qkv, c, qkv, 2 * c, m, h, w, splits, shifted)
                    self._conv(L["merge"], m, 0, q, 0)
                    n1 = L["norm1"]
                    if not ffn:      # a = a + norm1(merge(msg)), and the same value into the FFN's concat slot [a | .] of the cross part
                        self._c("vfi_layernorm_add", _p(q), c, c, B * h * w, _p(n1[0]), _p(n1[1]), _p(a), c, _p(a), c, _p(cat), 2 * c)
                    else:            # a = a + norm2(mlp(cat(a, norm1(merge(msg)))))
                        self._c("vfi_layernorm", _p(q), c, c, B * h * w, _p(n1[0]), _p(n1[1]), _p(cat, c), 2 * c)
                        self._conv(L["mlp0"], cat, 0, hid, 0, act=5)         # Linear + nn.GELU() in the conv's epilogue
                        self._conv(L["mlp2"], hid, 0, q, 0)
                        self._c("vfi_layernorm_add", _p(q), c, c, B * h * w, _p(L["norm2"][0]), _p(L["norm2"][1]), _p(a), c, _p(a), c, None, 0)

    def _add_position(self, t, splits):
        B, h, w, c = t.shape
        pos = self._const(("pos", B, h, w, c, splits),
                          lambda: sine_position(c, h // splits, w // splits).repeat(splits, splits, 1)[None].repeat(B, 1, 1, 1))
        self._ax(t, 0, pos, 0, t, 0, c)

    # ---- Model.reuse ------------------------------------------------------------------------------------------------
    fork_stages = True      # union head: IFNet 4.6 beside the splats on the engine's side stream (tests A/B it)

    def lone_pair(self, on):
        """lanes.tell_lone_pair: fork only when this is the only pair in flight"""
        self.fork_stages = bool(on)

    def prepare(self, frame0, frame1):
        assert frame1.shape == frame0.shape and frame0.shape[2] >= 3 and frame0.is_contiguous() and frame1.is_contiguous()
        # one HIP graph per frame shape (opsengine._replayable): the ~1000 launches of Model.reuse replay without the interpreter; the
        # record of what render() needs (tensors at the root of the pool: same addresses on every call) belongs to the graph
        self.prepared = self._replayable(("prepare",) + tuple(frame0.shape), (frame0, frame1), (), self._prepare)
        return self.prepared

    def _prepare(self, frame0, frame1):
        H, W, Cc = frame0.shape
        Hp, Wp = ((H - 1) // 64 + 1) * 64, ((W - 1) // 64 + 1) * 64
        Hh, Wh = Hp // 2, Wp // 2
        # what render() needs stays allocated (root of the pool); everything else lives in scopes and is recycled
        img = self._t("img", 2, Hp, Wp, 8)
        himg, flows, metric = self._t("himg", 2, Hh, Wh, 8), self._t("flows", 2, Hh, Wh, 2), self._t("metric", 1, Hh, Wh, 8)
        for i, f in enumerate((frame0, frame1)):
            self._c("vfi_pad_rgb", f.data_ptr(), Cc, H, W, _p(img[i]), 8, Hp, Wp)
        # FeatureNet on both frames
        feats, x, cin = [], img, 3
        for k, (a, first, second) in enumerate(self.fe):
            n, h, w, _ = x.shape
            f = self._t(f"feat{k}", 2, h // 2, w // 2, first["cout"])
            with self._scope():
                t = self._t(f"fe_in{k}", 2, h, w, _cs(cin))
                self._prelu(x, 0, t, 0, cin, a)
                m = self._t(f"fe_mid{k}", 2, h // 2, w // 2, first["cout"])
                self._conv(first, t, 0, m, 0)
                self._conv(second, m, 0, f, 0)
            feats.append(f)
            x, cin = f, first["cout"]
        self._resize(img, 0, himg, 0, 3)
        with self._scope():
            self._gmflow(himg, flows)
        # MetricNet
        with self._scope():
            mi = self._t("m_in", 1, Hh, Wh, 16)
            self._c("vfi_gmfss_metric_inputs", _p(himg[0]), _p(himg[1]), 8, _p(flows[0]), _p(flows[1]), 2, _p(mi), 16, Hh, Wh)
            feat, tmp, nxt = (self._t("m_" + n, 1, Hh, Wh, 64) for n in ("feat", "tmp", "nxt"))
            self._conv(self.m_in, mi, 0, feat, 0)
            for slope, conv in self.m_net:
                self._prelu(feat, 0, tmp, 0, 64, slope)
                self._conv(conv, tmp, 0, nxt, 0, res=feat)
                feat, nxt = nxt, feat
            self._prelu(feat, 0, tmp, 0, 64, self.m_out[0])
            self._conv(self.m_out[1], tmp, 0, metric, 0)
            self._c("vfi_tanh_scale", _p(metric), 8, 2, Hh * Wh, 10.0)
        self.prepared = dict(H=H, W=W, Hp=Hp, Wp=Wp, img=img, himg=himg, feats=feats, flows=flows, metric=metric)
        return self.prepared

    def _gmflow(self, himg, flows):
        """GMFlow.forward (:1262-1372) for both directions -> flows [2, Hh, Wh, 2] (index 0 = flow01, 1 = flow10)"""
        _, Hh, Wh, _ = himg.shape
        h4, w4 = Hh // 4, Wh // 4
        fhi, flo = self._t("f_hi", 2, h4, w4, 128), self._t("f_lo", 2, h4 // 2, w4 // 2, 128)
        with self._scope():
            self._backbone(himg, fhi, flo)
        flow8p = self._t("s0_flowp", 2, h4 // 2, w4 // 2, 2)
        with self._scope():
            self._match_global(flo, flow8p)
        with self._scope():
            self._refine_local(fhi, flow8p, flows)
        return flows

    def _backbone(self, himg, fhi, flo):
        """CNNEncoder (:218-272) + the trident convs: features at 1/4 (fhi) and 1/8 (flo) of the half-resolution pair"""
        _, Hh, Wh, _ = himg.shape
        nimg = self._t("nimg", 2, Hh, Wh, 8)
        mean, std = (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD)
        self._c("vfi_normalize_channels", _p(himg), 8, _p(nimg), 8, 3, 2 * Hh * Wh, mean, std)
        h2, w2 = Hh // 2, Wh // 2
        c1 = self._t("bb_c1", 2, h2, w2, 64)
        wt, bs, sl = self.bb_head
        self._c("vfi_conv7x7s2_prelu", _p(nimg), 8, _p(wt), _p(bs), _p(sl), 64, _p(c1), 64, 2, Hh, Wh)   # bias 0, slope 1: plain conv
        self._instnorm(c1, 64, True, None, False, c1)
        x, cin = c1, 64
        for name in ("layer1", "layer2", "layer3"):
            for blk in (0, 1):
                x = self._res_block((name, blk), x, cin)
                cin = x.shape[-1]
        h4, w4 = x.shape[1:3]
        assert (h4, w4) == tuple(fhi.shape[1:3])
        f4 = self._t("bb_f4", 2, h4, w4, 128)
        self._conv(self.bb_conv2, x, 0, f4, 0)
        self._conv(self.trident[0], f4, 0, fhi, 0)
        self._conv(self.trident[1], f4, 0, flo, 0)

    def _match_global(self, flo, flow8p):
        """scale 0 (:1293-1335 with attn_splits 2, global correlation, global propagation) -> flow8p at 1/8"""
        _, h8, w8, _ = flo.shape
        # global matching at 1/8, both directions from one transformer pass
        L = h8 * w8
        a = self._t("s0_a", 2, h8, w8, 128)
        self._ax(flo, 0, None, 0, a, 0, 128)
        self._add_position(a, 2)
        self._transformer(a, 2)
        o = self._t("s0_o", 2, h8, w8, 128)
        self._pair_swap(a, o)
        grid = self._const(("grid", h8, w8), lambda: torch.stack(torch.meshgrid(torch.arange(h8), torch.arange(w8), indexing="ij")[::-1], -1)
                           .float()[None].repeat(2, 1, 1, 1))
        flow8 = self._t("s0_flow", 2, h8, w8, 2)
        # global matching: softmax over ALL target pixels, expected coordinate (v = the pixel grid, 2 channels)
        self._c("vfi_attention", _p(a), 128, _p(o), 128, _p(grid), 2, _p(flow8), 2, 2, L, L, 128, 2, 1.0 / 128 ** 0.5, None, 0)
        self._ax(flow8, 0, grid, 0, flow8, 0, 2, 1.0, -1.0)
        # propagation: key projected from the PROJECTED query (:728-735)
        q, k = self._t("s0_q", 2, h8, w8, 128), self._t("s0_k", 2, h8, w8, 128)
        self._conv(self.prop_q, a, 0, q, 0)
        self._conv(self.prop_k, q, 0, k, 0)
        self._c("vfi_attention", _p(q), 128, _p(k), 128, _p(flow8), 2, _p(flow8p), 2, 2, L, L, 128, 2, 1.0 / 128 ** 0.5, None, 0)

    def _refine_local(self, fhi, flow8p, flows):
        """scale 1 (:1293-1372 with attn_splits 8, local correlation radius 4, local propagation, convex up-sampling)"""
        _, h4, w4, _ = fhi.shape
        h8, w8 = h4 // 2, w4 // 2
        Hh, Wh = flows.shape[1:3]
        # local refinement at 1/4, one batch entry pair per direction
        flow4 = self._t("s1_flow", 2, h4, w4, 2)
        self._c("vfi_resize_bilinear_ac", _p(flow8p), 2, _p(flow4), 2, 2, h8, w8, h4, w4, 2, 2.0)
        fsw = self._t("s1_fsw", 2, h4, w4, 128)
        self._pair_swap(fhi, fsw)                                     # target image of each direction
        b = self._t("s1_a", 4, h4, w4, 128)                           # [src d0, tgt d0, src d1, tgt d1]
        for d in (0, 1):
            self._ax(fhi[d:d + 1], 0, None, 0, b[2 * d:2 * d + 1], 0, 128)
            self._c("vfi_flow_sample", _p(fsw[d]), 128, _p(flow4[d]), 2, _p(b[2 * d + 1]), 128, 1, h4, w4, 128)
        self._add_position(b, 8)
        self._transformer(b, 8)
        q1, k1 = self._t("s1_q", 1, h4, w4, 128), self._t("s1_k", 1, h4, w4, 128)
        flow4p = self._t("s1_flowp", 2, h4, w4, 2)
        cat = self._t("s1_cat", 2, h4, w4, 136)
        for d in (0, 1):
            src = b[2 * d:2 * d + 1]
            self._c("vfi_local_match", _p(src), 128, _p(b[2 * d + 1]), 128, _p(flow4[d]), 2, 1, h4, w4, 128, 4)
            self._conv(self.prop_q, src, 0, q1, 0)
            self._conv(self.prop_k, src, 0, k1, 0)
            self._c("vfi_local_propagate", _p(q1), 128, _p(k1), 128, _p(flow4[d]), 2, _p(flow4p[d]), 2, 1, h4, w4, 128, 1)
            self._ax(src, 0, None, 0, cat[d:d + 1], 0, 128)
            self._ax(flow4p[d:d + 1], 0, None, 0, cat[d:d + 1], 128, 2)
        u = self._t("s1_up", 2, h4, w4, 256)
        self._conv(self.up0, cat, 0, u, 0, act=1, slope=0.0)          # ReLU
        msk = self._t("s1_mask", 2, h4, w4, 144)
        self._conv(self.up2, u, 0, msk, 0)
        assert (Hh, Wh) == (4 * h4, 4 * w4)
        self._c("vfi_convex_upsample", _p(msk), 144, _p(flow4p), 2, _p(flows), 2, 2, h4, w4, 4)

    # ---- Model.inference --------------------------------------------------------------------------------------------
    def render(self, t, out):
        P = self.prepared
        assert P is not None, "prepare() first"
        self._replayable(("render", P["H"], P["W"], float(t), self.fork_stages), (), (out,), lambda o: self._render(float(t), o))
        return out

    def _render(self, t, out):
        P = self.prepared
        H, W, Hp, Wp = P["H"], P["W"], P["Hp"], P["Wp"]
        Hh, Wh = Hp // 2, Wp // 2
        t = float(t)
        himg, flows, metric, feats = P["himg"], P["flows"], P["metric"], P["feats"]
        with self._scope():
            ft, zt = self._t("ft", 2, Hh, Wh, 2), self._t("zt", 2, Hh, Wh, 1)
            for d, tt in ((0, t), (1, 1 - t)):
                self._ax(flows[d:d + 1], 0, None, 0, ft[d:d + 1], 0, 2, tt)          # F_t = t * flow01, (1 - t) * flow10
                self._ax(metric, d, None, 0, zt[d:d + 1], 0, 1, tt)                  # Z_t
            g_in = [self._t("g_in0", 1, Hh, Wh, 16), self._t("g_in1", 1, Hh, Wh, 128), self._t("g_in2", 1, Hh // 2, Wh // 2, 256),
                    self._t("g_in3", 1, Hh // 4, Wh // 4, 384)]
            def splats():
                for lvl in range(3):
                    s = 1 << lvl
                    h, w = Hh // s, Wh // s
                    if lvl == 0:
                        fl, zl = ft, zt
                    else:   # F.interpolate(F_t, 1/s) * (1/s), F.interpolate(Z_t, 1/s)
                        fl, zl = self._t(f"ft{lvl}", 2, h, w, 2), self._t(f"zt{lvl}", 2, h, w, 1)
                        self._resize(ft, 0, fl, 0, 2, 1.0 / s)
                        self._resize(zt, 0, zl, 0, 1)
                    c = feats[lvl].shape[-1]
                    for d in (0, 1):
                        if lvl == 0:   # union head: (I1t, rife, I2t); base head: (img0, I1t, I2t, img1)
                            self._splat(himg[d:d + 1], 3, zl[d:d + 1], fl[d:d + 1], g_in[0], 6 * d if self.union else 3 + 3 * d)
                        self._splat(feats[lvl][d:d + 1], c, zl[d:d + 1], fl[d:d + 1], g_in[lvl + 1], c * d)

            cur, side = self._fork() if (self.union and self.fork_stages) else (None, None)
            if side is not None:
                # r6: the union head's IFNet 4.6 pass reads the half-resolution frames only: it runs on the engine's side stream beside the
                # 14 splats (small, latency-bound launches); its temporaries are held until the join
                with self._hold() as hold:
                    with torch.cuda.stream(side):
                        hold["active"] = True
                        self._ifnet46(himg, t, g_in[0], 3)
                        hold["active"] = False
                    splats()
                    self._join(cur, side)
            else:
                splats()
                if self.union:
                    self._ifnet46(himg, t, g_in[0], 3)
            if not self.union:
                self._ax(himg[0:1], 0, None, 0, g_in[0], 0, 3)
                self._ax(himg[1:2], 0, None, 0, g_in[0], 9, 3)
            y = self._gridnet(g_in)
            self._c("vfi_clamp_crop", _p(y), y.shape[-1], Hp, Wp, out.data_ptr(), H, W, 3)
        return out

    def _splat(self, x, c, z, flow, dst, doff):
        """softsplat(x, flow, z, "soft") -> dst[..., doff:doff+c]"""
        _, h, w, _ = x.shape
        with self._scope():
            pre, fo, s = self._t("sp_pre", h * w, c + 1), self._t("sp_flow", h * w, 2), self._t("sp_out", h * w, c + 1)
            self._c("vfi_splat_prep", _p(x), x.shape[-1], _p(z), z.shape[-1], _p(flow), flow.shape[-1], _p(pre), _p(fo), c, h * w, 1.0, 1.0)
            self._c("vfi_softsplat_sum", _p(pre), _p(fo), _p(s), 1, h, w, c + 1)
            self._c("vfi_splat_normalize", _p(s), _p(dst, doff), dst.shape[-1], c, h * w)

    def _ifnet46(self, himg, t, dst, doff):
        """IFNet("4.6").forward (rife_arch.py:465-732) on the half-resolution pair -> dst[..., doff:doff+3]"""
        _, Hh, Wh, _ = himg.shape
        Hq, Wq = ((Hh - 1) // 64 + 1) * 64, ((Wh - 1) // 64 + 1) * 64
        with self._scope():
            x7 = self._t("r_x7", 1, Hq, Wq, 8)          # clamp(img0), clamp(img1), t, 0
            self._c("vfi_rife40_prep", _p(himg[0]), _p(himg[1]), 8, Hh, Wh, t, _p(x7), Hq, Wq)
            xin = self._t("r_xin", 1, Hq, Wq, 8)        # warped img0, warped img1, t, mask
            self._ax(x7, 6, None, 0, xin, 6, 1)
            flow, df, dm = self._t("r_flow", 1, Hq, Wq, 4), self._t("r_df", 1, Hq, Wq, 4), self._t("r_dm", 1, Hq, Wq, 1)
            for i, scale in enumerate((8, 4, 2, 1)):
                B = self.rife[i]
                c = B["c"]
                hs, ws = Hq // scale, Wq // scale
                xs = self._t(f"r_xs{i}", 1, hs, ws, 8 if i == 0 else 16)
                self._resize(x7 if i == 0 else xin, 0, xs, 0, 7 if i == 0 else 8)
                if i > 0:
                    self._resize(flow, 0, xs, 8, 4, 1.0 / scale)
                a = self._t(f"r_a{i}", 1, hs // 2, ws // 2, c // 2)
                p, q = self._t(f"r_p{i}", 1, hs // 4, ws // 4, c), self._t(f"r_q{i}", 1, hs // 4, ws // 4, c)
                self._conv(B["c00"], xs, 0, a, 0, act=1, slope=0.2)
                self._conv(B["c01"], a, 0, p, 0, act=1, slope=0.2)
                cur, nxt = p, q
                for L in B["res"]:                      # lrelu(conv(x) * beta + x)
                    self._conv(L, cur, 0, nxt, 0, act=1, slope=0.2, res=cur)
                    cur, nxt = nxt, cur
                t24 = self._t(f"r_t24{i}", 1, hs // 2, ws // 2, 24)
                self._conv(B["last"], cur, 0, t24, 0)
                t6 = self._t(f"r_t6{i}", 1, hs, ws, 8)
                self._c("vfi_pixel_shuffle2", _p(t24), 24, _p(t6), 8, 1, hs // 2, ws // 2, 6)
                self._resize(t6, 0, df, 0, 4, float(scale))
                self._resize(t6, 4, dm, 0, 1)
                if i == 0:
                    self._ax(df, 0, None, 0, flow, 0, 4)
                    self._ax(dm, 0, None, 0, xin, 7, 1)
                else:
                    self._ax(flow, 0, df, 0, flow, 0, 4)
                    self._ax(xin, 7, dm, 0, xin, 7, 1)
                for k in (0, 1):                        # warped_img0 = warp(img0, flow[:, :2]), warped_img1 = warp(img1, flow[:, 2:4])
                    self._c("vfi_warp_rife", _p(x7, 3 * k), 8, _p(flow, 2 * k), 4, _p(xin, 3 * k), 8, 1, Hq, Wq, 3)
            rife = self._t("r_out", 1, Hh, Wh, 3)
            self._c("vfi_rife40_output", _p(xin), 8, _p(xin, 7), 8, None, 0, _p(rife), 1, Hq, Wq, Hh, Wh)
            self._ax(rife, 0, None, 0, dst, doff, 3)

    def _gridnet(self, g_in):
        """GridNet.forward (:1639-1688) -> [1, Hp, Wp, 8] (3 channels used).  Every state is dropped after its last reader: of the
        24 states (2.8 GiB at 1080p) at most a handful are alive at once."""
        x, x1, x2, x3 = g_in
        _, h, w, _ = x.shape
        T = lambda name, s, c: self._t("gn_" + name, 1, h // s, w // s, c)   # noqa: E731
        P, D = self._pair, self._drop
        a = T("h0", 1, 64)
        P(self.gn["reshead0"], x, 0, 9 if self.union else 12, a, 0, "h0")
        x00 = T("x00", 1, 64)
        P(self.gn["reshead1"], x1, 0, 128, x00, 0, "h1", res=a)
        D(a)
        x01 = T("x01", 1, 64)
        P(self.gn["res01"], x00, 0, 64, x01, 0, "r0", res=x00)
        b = T("h2", 2, 128)
        P(self.gn["reshead2"], x2, 0, 256, b, 0, "h2")
        x10 = T("x10", 2, 128)
        P(self.gn["down10"], x00, 0, 64, x10, 0, "d1", res=b)
        D(b, x00)
        c = T("h3", 4, 192)
        P(self.gn["reshead3"], x3, 0, 384, c, 0, "h3")
        x20 = T("x20", 4, 192)
        P(self.gn["down20"], x10, 0, 128, x20, 0, "d2", res=c)
        D(c)
        r11 = T("r11", 2, 128)
        P(self.gn["res11"], x10, 0, 128, r11, 0, "r1", res=x10)
        D(x10)
        x11 = T("x11", 2, 128)
        P(self.gn["down11"], x01, 0, 64, x11, 0, "d1", res=r11)          # residual_11 + downsample_11 (the sum commutes)
        D(r11)
        r21 = T("r21", 4, 192)
        P(self.gn["res21"], x20, 0, 192, r21, 0, "r2", res=x20)
        D(x20)
        x21 = T("x21", 4, 192)
        P(self.gn["down21"], x11, 0, 128, x21, 0, "d2", res=r21)
        D(r21)
        x24 = T("x24", 4, 192)
        P(self.gn["res24"], x21, 0, 192, x24, 0, "r2", res=x21)
        D(x21)
        x25 = T("x25", 4, 192)
        P(self.gn["res25"], x24, 0, 192, x25, 0, "r2", res=x24)
        r14 = T("r14", 2, 128)
        P(self.gn["res14"], x11, 0, 128, r14, 0, "r1", res=x11)
        D(x11)
        x14 = T("x14", 2, 128)
        P(self.gn["up14"], x24, 0, 192, x14, 0, "u1", res=r14)
        D(r14, x24)
        r04 = T("r04", 1, 64)
        P(self.gn["res04"], x01, 0, 64, r04, 0, "r0", res=x01)
        D(x01)
        x04 = T("x04", 1, 64)
        P(self.gn["up04"], x14, 0, 128, x04, 0, "u0", res=r04)
        D(r04)
        r15 = T("r15", 2, 128)
        P(self.gn["res15"], x14, 0, 128, r15, 0, "r1", res=x14)
        D(x14)
        x15 = T("x15", 2, 128)
        P(self.gn["up15"], x25, 0, 192, x15, 0, "u1", res=r15)
        D(r15, x25)
        r05 = T("r05", 1, 64)
        P(self.gn["res05"], x04, 0, 64, r05, 0, "r0", res=x04)
        D(x04)
        x05 = T("x05", 1, 64)
        P(self.gn["up05"], x15, 0, 128, x05, 0, "u0", res=r05)
        D(r05, x15)
        t0, t1 = T("t0", 1, 64), T("t1", 1, 256)
        self._conv(self.tail[0], x05, 0, t0, 0)
        D(x05)
        self._conv(self.tail[1], t0, 0, t1, 0)
        D(t0)
        ps = self._t("gn_ps", 1, 2 * h, 2 * w, 64)
        self._c("vfi_pixel_shuffle2", _p(t1), 256, _p(ps), 64, 1, h, w, 64)
        D(t1)
        y = self._t("gn_y", 1, 2 * h, 2 * w, 8)
        self._conv(self.tail[2], ps, 0, y, 0)
        D(ps)
        return y

    def forward(self, frame0, frame1, t, out):
        self.prepare(frame0, frame1)
        return self.render(t, out)


def _load(path):
    sd = torch.load(path, map_location="cpu", weights_only=False)
    return sd["state_dict"] if isinstance(sd, dict) and "state_dict" in sd else sd


class GMFSS_Fortuna_VFI:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (list(CKPTS_PATH_CONFIG.keys()),),
                "frames": ("IMAGE",),
                "clear_cache_after_n_frames": ("INT", {"default": 10, "min": 1, "max": 1000}),
                "multiplier": ("INT", {"default": 2, "min": 2, "max": 1000}),
            },
            "optional": {"optional_interpolation_states": ("INTERPOLATION_STATES",)},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "vfi"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def vfi(self, ckpt_name: typing.AnyStr, frames: torch.Tensor, clear_cache_after_n_frames=10, multiplier: typing.SupportsInt = 2,
            optional_interpolation_states: InterpolationStateList = None, **kwargs):
        from .ckpt import load_file_from_github_release
        from .m2m import run_plan

        assert len(frames) >= 2, f"VFI model GMFSS Fortuna requires at least 2 frames to work with, only found {frames.shape[0]}."
        if ckpt_name not in CKPTS_PATH_CONFIG:
            raise KeyError(ckpt_name)
        from .ckpt import begin_call, cached_engine, end_call
        from .lanes import lane_set
        paths = {part: load_file_from_github_release(*loc) for part, loc in CKPTS_PATH_CONFIG[ckpt_name].items()}

        def build():
            sds = {part: _load(path) for part, path in paths.items()}
            return lane_set("gmfss", lambda: GMFSSEngine(sds))
        # (the reference rebuilds the model on every call, gmfss_fortuna/__init__.py:129-130.  Here the packed weights stay between calls —
        # a constructor is 60-90 ms per lane — keyed by the variant and its fusion-net file; see ckpt.cached_engine)
        engine, cached = cached_engine(MODEL_TYPE + ":" + ckpt_name, paths["fusionnet"], build)
        try:
            begin_call(engine, frames.shape[1:3])
            plan, tasks = generic_output_plan(len(frames), multiplier, optional_interpolation_states)
            return (run_plan(engine, frames, plan, tasks, name="GMFSS Fortuna VFI"),)
        finally:
            torch.cuda.synchronize(engine.device)
            end_call(engine, cached)      # (the workspace and the captured graphs stay for the next call of this frame shape: ckpt.KEEP_WORKSPACE_BYTES)
