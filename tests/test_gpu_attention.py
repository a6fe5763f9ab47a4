"""-m gpu: vfi_attention (csrc/attention.hip, flash-style fp32-MFMA attention) against the plain torch formulation
softmax(alpha q k^T + mask) v in float64 — GMFlow's window attention (GMFSS_Fortuna_union_arch.py:367-436), global matching
(:806-843) and global flow propagation (:708-745); lengths that are not multiples of the 32-key block or the 128-query
workgroup, the shifted-window label mask, 2-channel values."""
import ctypes as C

import pytest
import torch

from gpu_util import describe_diff

pytestmark = pytest.mark.gpu


def _ref(q, k, v, alpha, labels):
    sc = torch.matmul(q.double(), k.double().transpose(1, 2)) * alpha
    if labels is not None:
        nb, period = q.shape[0], labels.shape[0]
        m = torch.where(labels[:, :, None] != labels[:, None, :], torch.tensor(-100.0, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64))
        sc = sc + m.repeat(nb // period, 1, 1)
    return torch.matmul(torch.softmax(sc, dim=-1), v.double()).float()


@pytest.mark.parametrize("nb,lq,lk,dv,period,gain", [(2, 64, 64, 128, 0, 1.0), (8, 510, 510, 128, 4, 1.0), (3, 2040, 2040, 128, 0, 1.0),
                                                     (2, 1000, 1000, 2, 0, 1.0), (2, 130, 97, 128, 0, 1.0), (1, 33, 31, 2, 0, 1.0),
                                                     (4, 510, 510, 128, 4, 6.0), (2, 777, 777, 2, 0, 8.0)])
def test_attention_matches_torch(hip_lib, nb, lq, lk, dv, period, gain):
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(nb * 1000 + lq + dv)
    q = torch.randn(nb, lq, 128, generator=g) * gain          # gain > 1: near one-hot softmax rows (the coherent GMFlow regime)
    k = torch.randn(nb, lk, 128, generator=g) * gain
    v = torch.randn(nb, lk, dv, generator=g) if dv == 128 else torch.rand(nb, lk, dv, generator=g) * 100
    labels = torch.randint(0, 3, (period, lk), generator=g, dtype=torch.int32) if period else None
    alpha = 1.0 / 128 ** 0.5
    want = _ref(q, k, v, alpha, labels)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    ld = labels.cuda() if labels is not None else None
    out = torch.full((nb, lq, dv), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_attention(qd.data_ptr(), 128, kd.data_ptr(), 128, vd.data_ptr(), dv, out.data_ptr(), dv, nb, lq, lk, 128, dv,
                                     alpha, ld.data_ptr() if ld is not None else None, period, None), "vfi_attention")
    torch.cuda.synchronize()
    got = out.cpu()
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"attention nb={nb} {lq}x{lk} dv={dv}", chan_last=False)


def test_attention_in_channel_windows(hip_lib):
    """q / k / v / out as windows of wider tensors (pixel strides larger than the channel counts)"""
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(5)
    big = torch.randn(2, 200, 512, generator=g).cuda()
    out = torch.full((2, 200, 160), float("nan"), device="cuda")
    q, k, v = big[..., 0:128], big[..., 128:256], big[..., 256:384]
    _lib.check(hip_lib.vfi_attention(q.data_ptr(), 512, k.data_ptr(), 512, v.data_ptr(), 512, out.data_ptr() + 16 * 4, 160, 2, 200, 200, 128, 128,
                                     0.1, None, 0, None), "vfi_attention")
    torch.cuda.synchronize()
    want = _ref(q.cpu(), k.cpu(), v.cpu(), 0.1, None)
    got = out.cpu()
    assert torch.isnan(got[..., :16]).all() and torch.isnan(got[..., 144:]).all(), "wrote outside its channel window"
    assert (got[..., 16:144] - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    with pytest.raises(RuntimeError, match="head dimension"):
        _lib.check(hip_lib.vfi_attention(q.data_ptr(), 512, k.data_ptr(), 512, v.data_ptr(), 512, out.data_ptr(), 160, 2, 200, 200, 64, 128, 0.1,
                                         None, 0, None), "vfi_attention")
