"""TEST DOUBLE of the C ABI (include/vfi_hip.h) for orchestration tests on the CPU — test infrastructure only.

The GMFSS engine (comfyui-frame-interpolation_amd/gmfss.py) talks to its backend through raw pointers, like any client of
libvfi_hip.so.  This backend serves those calls on HOST memory so that the engine's orchestration (layer wiring, channel
windows, batch layout, schedule) can be checked against the oracle without a GPU:
  * the GMFSS kernels of csrc/gmfss_ops.hip: the real per-element bodies, through tests/hostcheck (same entry points);
  * the older entry points the engine also uses (layer objects, resize, axpby, warps, the summation splat, ...): small torch
    restatements over strided views of the pointed-to memory, following the documented semantics of include/vfi_hip.h.
The package never imports this module; engines take it only through the explicit ``_test_backend`` argument, and the product
constructors raise without a GPU.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

import hostcheck
from oracle import m2m_oracle, rife_oracle


def view(ptr, n, h, w, cs, c):
    """torch view [n,h,w,c] of float32 memory at ``ptr`` laid out NHWC with pixel stride ``cs``"""
    span = (n * h * w - 1) * cs + c
    arr = np.ctypeslib.as_array((C.c_float * span).from_address(int(ptr)))
    return torch.as_strided(torch.from_numpy(arr), (n, h, w, c), (h * w * cs, w * cs, cs, 1))


def host_array(ptr, count, ctype=C.c_float):
    if hasattr(ptr, "_length_"):   # a ctypes array passed directly
        return torch.tensor(list(ptr))
    return torch.from_numpy(np.ctypeslib.as_array((ctype * count).from_address(int(ptr)))).clone()


def nchw(x):
    return x.permute(0, 3, 1, 2)


# ---- host-side contract of the MFMA convolution launcher ---------------------------------------------------------------------
# Mirror (test infrastructure) of conv_pick_variant + the VFI_REQUIREs of launch_t / launch2_e in csrc/conv_mfma.hip and
# csrc/conv_mfma2.hip: which tile variant a layer call gets on the MI355X and the divisibility that variant demands.  The test
# double asserts it on every vfi_conv_forward_ex call, so a layer shape the real launcher would refuse ("Cin_p=72 not a multiple
# of the K chunk 16" — how IFRNet's first GPU run failed) is caught on the CPU.  Keep in sync with those two files.
_K2 = 32                                                   # kConv2Base
_V1 = {0: (16, 64), 1: (16, 96), 2: (16, 64), 3: (16, 96), 4: (16, 32), 5: (16, 32), 6: (16, 64), 7: (16, 128), 8: (8, 64), 9: (8, 96),
       10: (8, 32), 11: (8, 64), 12: (16, 32), 13: (16, 32)}                       # variant -> (K chunk, N tile)
_V2 = {0: (8, 64), 1: (8, 96), 2: (8, 64), 3: (8, 96), 4: (8, 128), 5: (8, 128), 6: (16, 64), 7: (8, 64), 8: (8, 64), 9: (8, 96), 10: (8, 32),
       11: (8, 32), 12: (8, 32), 13: (8, 32), 14: (8, 64), 15: (8, 64), 16: (8, 64), 17: (8, 64), 18: (8, 32), 19: (8, 32), 20: (8, 32), 21: (8, 64),
       22: (8, 64), 23: (32, 64), 24: (32, 128)}


def conv_variant(grouped, taps, stride, cin_p, cout_p, px, out_mode=0, n_cus=256):
    if not grouped and stride == 2 and taps == 4:
        return _K2 + (21 if cout_p % 64 == 0 else 20)
    if not grouped and stride == 1 and taps != 9:
        big = px * (cout_p // 32) >= 256 * 4 * n_cus
        if taps == 1 and cout_p % 64 == 0 and cin_p % 32 == 0:
            return _K2 + (24 if (cin_p >= 512 and cout_p % 128 == 0) else 23)
        if cout_p % 64 == 0:
            return _K2 + ((14 if big else 15) if taps == 4 else (16 if big else 17))
        return _K2 + (19 if taps == 4 else 18)
    if grouped:
        if out_mode == 1 and px >= 250000:
            return _K2 + 11
        return 13 if cin_p % 16 == 0 else _K2 + 11
    n3, n2 = cout_p % 96 == 0, cout_p % 64 == 0
    if stride == 2:
        return _K2 + 7 if n2 else (9 if n3 else 10)
    if n2:
        return _K2 + (0 if cout_p == 64 else 2)
    if n3:
        return _K2 + 3 if (px >= 100000 or cin_p % 16) else 4
    return 4 if (cin_p % 16 == 0 and px < 20000) else _K2 + 13


def assert_conv_contract(L, n, hin, win, in_cs):
    grouped = L["kind"] == 1
    taps = 4 if grouped else L["k"] * L["k"]
    cout_p = (L["cout"] + 31) // 32 * 32
    hout, wout = (hin, win) if grouped else (hin // L["stride"], win // L["stride"])
    v = conv_variant(grouped, taps, 1 if grouped else L["stride"], L["cin_phys"], cout_p, n * hout * wout, 2 if grouped else 0)
    ck, bn = _V2[v - _K2] if v >= _K2 else _V1[v]
    what = f"conv k={L['k']} s={L['stride']} kind={L['kind']} {L['cin_phys']}->{L['cout']} on {n}x{hin}x{win}: variant {v}"
    assert L["cin_phys"] % ck == 0, f"{what}: Cin_p not a multiple of the K chunk {ck}"
    assert cout_p % bn == 0, f"{what}: Cout_p={cout_p} not a multiple of the N tile {bn}"
    assert hin * win * in_cs * 4 < 0x7FFFFFFF, f"{what}: image larger than 2 GiB"


class EmuLib:
    def __init__(self):
        self._hc = hostcheck.load()
        self._layers, self._next = {}, 1
        self.calls = {}

    def __getattr__(self, name):   # every entry point without an override below: the host-side build of the real kernels
        fn = getattr(self._hc, name)

        def counted(*a):
            self.calls[name] = self.calls.get(name, 0) + 1
            return fn(*a)

        return counted

    # ---- flash attention (csrc/attention.hip is an MFMA kernel without a host body): torch restatement of its contract ---
    def vfi_attention(self, q_ptr, q_cs, k_ptr, k_cs, v_ptr, v_cs, out_ptr, out_cs, nb, lq, lk, c, dv, alpha, labels_ptr, period, stream=None):
        assert c == 128 and (dv == 128 or 1 <= dv <= 32) and q_cs >= c and k_cs >= c and v_cs >= dv and out_cs >= dv
        assert int(q_ptr) % 16 == 0 and int(k_ptr) % 16 == 0 and (not labels_ptr or (lq == lk and period > 0))
        q = view(q_ptr, nb, 1, lq, q_cs, c)[:, 0]
        k = view(k_ptr, nb, 1, lk, k_cs, c)[:, 0]
        v = view(v_ptr, nb, 1, lk, v_cs, dv)[:, 0]
        sc = torch.matmul(q, k.transpose(1, 2)) * alpha
        if labels_ptr:
            lab = host_array(labels_ptr, period * lk, C.c_int).view(period, lk)
            m = torch.where(lab[:, :, None] != lab[:, None, :], torch.tensor(-100.0), torch.tensor(0.0))
            sc = sc + m.repeat(nb // period, 1, 1)
        view(out_ptr, nb, 1, lq, out_cs, dv)[:, 0] = torch.matmul(torch.softmax(sc, dim=-1), v)
        self.calls["vfi_attention"] = self.calls.get("vfi_attention", 0) + 1
        return 0

    def vfi_window_attention(self, q_ptr, q_cs, k_ptr, k_cs, v_ptr, v_cs, out_ptr, out_cs, B, h, w, splits, sh, sw, c, alpha, labels_ptr, stream=None):
        """roll, split into windows, attention, merge, roll back — the torch statement of csrc/attention.hip's window mode"""
        assert c == 128 and h % splits == 0 and w % splits == 0 and 0 <= sh < h // splits and 0 <= sw < w // splits
        assert out_ptr not in (q_ptr, k_ptr, v_ptr)
        wh, ww, K = h // splits, w // splits, splits

        def windows(ptr, cs):
            t = torch.roll(view(ptr, B, h, w, cs, c), shifts=(-sh, -sw), dims=(1, 2))
            return t.view(B, K, wh, K, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(B * K * K, wh * ww, c)

        q, k, v = windows(q_ptr, q_cs), windows(k_ptr, k_cs), windows(v_ptr, v_cs)
        sc = torch.matmul(q, k.transpose(1, 2)) * alpha
        if labels_ptr:
            lab = host_array(labels_ptr, K * K * wh * ww, C.c_int).view(K * K, wh * ww)
            m = (lab[:, :, None] != lab[:, None, :]).float() * -100.0
            sc = sc + m.repeat(B, 1, 1)
        o = torch.matmul(torch.softmax(sc, dim=-1), v).view(B, K, K, wh, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, c)
        view(out_ptr, B, h, w, out_cs, c)[:] = torch.roll(o, shifts=(sh, sw), dims=(1, 2))
        self.calls["vfi_window_attention"] = self.calls.get("vfi_window_attention", 0) + 1
        return 0

    # ---- layer objects (vfi_conv_create_ex / vfi_conv_forward_ex) -----------------------------------------------------
    def vfi_conv_create_ex(self, kind, w_ptr, b_ptr, cout, cin, k, stride, pad_mode, chan_map, cin_phys, prelu_ptr):
        assert pad_mode == 0 and cin_phys % 8 == 0 and cin_phys >= cin
        assert (kind == 0 and ((k == 3 and stride in (1, 2)) or (k == 1 and stride == 1) or (k == 2 and stride == 2))) or (kind == 1 and k == 4 and stride == 2)
        shape = (cout, cin, k, k) if kind == 0 else (cin, cout, 4, 4)
        w = host_array(w_ptr, int(np.prod(shape))).view(shape)
        b = host_array(b_ptr, cout) if b_ptr else torch.zeros(cout)
        cm = list(chan_map) if chan_map is not None else list(range(cin))
        assert all(0 <= c < cin_phys for c in cm)
        pr = host_array(prelu_ptr, cout) if prelu_ptr else None
        h = self._next
        self._next += 1
        self._layers[h] = dict(kind=kind, w=w, b=b, cout=cout, cin=cin, k=k, stride=stride, cm=cm, cin_phys=cin_phys, prelu=pr)
        return h

    def vfi_conv_destroy(self, h):
        self._layers.pop(h, None)

    def vfi_conv_forward_ex(self, h, in_ptr, in_cs, hin, win, out_ptr, out_cs, n, act, slope, post_scale, post_shift, res_ptr, res_cs, stream):
        L = self._layers[h]
        assert in_cs >= L["cin_phys"] and in_cs % 4 == 0 and int(in_ptr) % 16 == 0, "input window must hold Cin_phys channels, 16-byte aligned"
        assert act != 3 or L["prelu"] is not None
        assert_conv_contract(L, n, hin, win, in_cs)
        x = nchw(view(in_ptr, n, hin, win, in_cs, L["cin_phys"])[..., L["cm"]])
        if L["kind"] == 1:
            assert not res_ptr
            y = F.conv_transpose2d(x, L["w"], L["b"], 2, 1)
        else:
            assert L["stride"] == 1 or (hin % 2 == 0 and win % 2 == 0)
            y = F.conv2d(x, L["w"], L["b"], L["stride"], L["k"] // 2 if L["k"] != 2 else 0)
        ho, wo = y.shape[2:]
        if res_ptr:
            y = y + nchw(view(res_ptr, n, ho, wo, res_cs, L["cout"]))
        if act == 1:
            y = F.leaky_relu(y, slope)
        elif act == 3:
            y = F.prelu(y, L["prelu"])
        elif act == 4:
            y = torch.sigmoid(y)
        elif act == 5:
            y = F.gelu(y)
        else:
            assert act == 0
        assert post_scale == 0.0
        view(out_ptr, n, ho, wo, out_cs, L["cout"]).copy_(y.permute(0, 2, 3, 1))
        self.calls["vfi_conv_forward_ex"] = self.calls.get("vfi_conv_forward_ex", 0) + 1
        return 0

    # ---- generic NHWC ops ------------------------------------------------------------------------------------------------
    def vfi_axpby(self, a_ptr, a_cs, b_ptr, b_cs, out_ptr, out_cs, px, c, alpha, beta, stream):
        v = view(a_ptr, 1, 1, px, a_cs, c) * np.float32(alpha)
        if b_ptr:
            v = v + view(b_ptr, 1, 1, px, b_cs, c) * np.float32(beta)
        view(out_ptr, 1, 1, px, out_cs, c).copy_(v)
        return 0

    def vfi_resize_bilinear(self, in_ptr, in_cs, out_ptr, out_cs, n, hin, win, hout, wout, c, mul, stream):
        x = nchw(view(in_ptr, n, hin, win, in_cs, c)) * np.float32(mul)
        y = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=False)
        view(out_ptr, n, hout, wout, out_cs, c).copy_(y.permute(0, 2, 3, 1))
        return 0

    def vfi_upsample_nearest(self, in_ptr, in_cs, out_ptr, out_cs, n, hin, win, hout, wout, c, stream):
        assert c % 4 == 0
        y = F.interpolate(nchw(view(in_ptr, n, hin, win, in_cs, c)), size=(hout, wout), mode="nearest")
        view(out_ptr, n, hout, wout, out_cs, c).copy_(y.permute(0, 2, 3, 1))
        return 0

    def vfi_conv7x7s2_prelu(self, in_ptr, in_cs, w_ptr, b_ptr, s_ptr, cout, out_ptr, out_cs, n, hin, win, stream):
        w = host_array(w_ptr, 7 * 7 * 3 * cout).view(7, 7, 3, cout).permute(3, 2, 0, 1)
        y = F.prelu(F.conv2d(nchw(view(in_ptr, n, hin, win, in_cs, 3)), w, host_array(b_ptr, cout), 2, 3), host_array(s_ptr, cout))
        view(out_ptr, n, y.shape[2], y.shape[3], out_cs, cout).copy_(y.permute(0, 2, 3, 1))
        return 0

    def vfi_softsplat_sum(self, in_ptr, flow_ptr, out_ptr, n, h, w, c, stream):
        x = nchw(view(in_ptr, n, h, w, c, c)).contiguous().numpy()
        f = nchw(view(flow_ptr, n, h, w, 2, 2)).contiguous().numpy()
        view(out_ptr, n, h, w, c, c).copy_(torch.from_numpy(m2m_oracle.softsplat_sum(x, f)).permute(0, 2, 3, 1))
        return 0

    # ---- RIFE arch 4.0 helpers re-used by IFNet 4.6 ---------------------------------------------------------------------------
    def vfi_rife40_prep(self, f0_ptr, f1_ptr, c, h, w, timestep, out_ptr, hp, wp, stream):
        out = view(out_ptr, 1, hp, wp, 8, 8)
        out.zero_()
        out[0, :h, :w, 0:3] = view(f0_ptr, 1, h, w, c, 3)[0].clamp(0, 1)
        out[0, :h, :w, 3:6] = view(f1_ptr, 1, h, w, c, 3)[0].clamp(0, 1)
        out[..., 6] = float(np.float32(timestep))
        return 0

    def vfi_warp_rife(self, in_ptr, in_cs, flow_ptr, flow_cs, out_ptr, out_cs, n, h, w, c, stream):
        y = rife_oracle.warp(nchw(view(in_ptr, n, h, w, in_cs, c)).contiguous(), nchw(view(flow_ptr, n, h, w, flow_cs, 2)).contiguous())
        view(out_ptr, n, h, w, out_cs, c).copy_(y.permute(0, 2, 3, 1))
        return 0

    def vfi_rife40_output(self, w01_ptr, w_cs, mask_ptr, m_cs, res_ptr, r_cs, out_ptr, b, hp, wp, h, w, stream):
        assert not res_ptr
        w01 = view(w01_ptr, b, hp, wp, w_cs, 6)
        m = torch.sigmoid(view(mask_ptr, b, hp, wp, m_cs, 1))
        y = (w01[..., 0:3] * m + w01[..., 3:6] * (1 - m)).clamp(0, 1)
        view(out_ptr, b, h, w, 3, 3).copy_(y[:, :h, :w])
        return 0


class EmuBackend:
    def __init__(self):
        self.lib = EmuLib()
        self.device = torch.device("cpu")

    @staticmethod
    def stream():
        return None

    @staticmethod
    def last_error():
        return "(emulated backend)"
