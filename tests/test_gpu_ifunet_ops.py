"""-m gpu: CBAM's reductions (IFUNet_arch.py:411-503) as the MI355X runs them — the cooperative kernels of csrc/ifunet_fast.hip
that vfi_channel_pool / vfi_cbam_gate / vfi_cbam_scale_compress dispatch to on the device — against their torch statement.
(The per-element bodies they replace are checked on the host by tests/test_ifunet_bodies_cpu.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c,cs,hw,strips", [(1, 32, 32, 1000, 64), (2, 96, 128, 777, 512), (1, 256, 256, 4321, 100), (1, 320, 320, 500, 64)])
def test_channel_pool(hip_lib, n, c, cs, hw, strips):
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(n + c + hw)
    x = torch.randn(n, hw, cs, generator=g) * 2 + 0.5
    xd = x.cuda()
    stats = torch.full((n, c, 2), float("nan"), device="cuda")
    ws = torch.zeros(n * strips * c * 3, device="cuda")
    _lib.check(hip_lib.vfi_channel_pool(xd.data_ptr(), cs, c, n, hw, stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, None), "vfi_channel_pool")
    torch.cuda.synchronize()
    got = stats.cpu()
    assert (got[..., 0] - x[..., :c].double().mean(1).float()).abs().max().item() <= 1e-6
    assert torch.equal(got[..., 1], x[..., :c].amax(1))


@pytest.mark.parametrize("n,c,r", [(1, 32, 2), (2, 256, 16), (1, 96, 6), (3, 512, 32)])
def test_cbam_gate(hip_lib, n, c, r):
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(c + r)
    stats = torch.randn(n, c, 2, generator=g)
    w1, b1 = torch.randn(r, c, generator=g) * 0.2, torch.randn(r, generator=g) * 0.1
    w2, b2 = torch.randn(c, r, generator=g) * 0.2, torch.randn(c, generator=g) * 0.1

    def mlp(v):
        return torch.relu(v.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()

    want = torch.sigmoid(mlp(stats[..., 0]) + mlp(stats[..., 1])).float()
    dev = [t.cuda() for t in (stats, w1, b1, w2, b2)]
    scale = torch.full((n, c), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_cbam_gate(*(t.data_ptr() for t in dev), c, r, n, scale.data_ptr(), None), "vfi_cbam_gate")
    torch.cuda.synchronize()
    assert (scale.cpu() - want).abs().max().item() <= 2e-6


@pytest.mark.parametrize("n,c,cs,hw", [(1, 32, 32, 1001), (2, 96, 96, 333), (1, 256, 256, 517), (1, 17, 24, 100), (1, 64, 72, 4097)])
def test_cbam_scale_compress(hip_lib, n, c, cs, hw):
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(c + hw)
    x = torch.randn(n, hw, cs, generator=g)
    scale = torch.rand(n, c, generator=g)
    xd, sd = x.cuda(), scale.cuda()
    xs = torch.full((n, hw, cs), float("nan"), device="cuda")
    comp = torch.full((n * hw, 2), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_cbam_scale_compress(xd.data_ptr(), cs, sd.data_ptr(), c, n, hw, xs.data_ptr(), cs, comp.data_ptr(), None),
               "vfi_cbam_scale_compress")
    torch.cuda.synchronize()
    want = x[..., :c] * scale[:, None, :]
    got = xs.cpu()
    assert torch.equal(got[..., :c], want) and (cs == c or torch.isnan(got[..., c:]).all())
    cp = comp.cpu().view(n, hw, 2)
    assert torch.equal(cp[..., 0], want.amax(-1))
    assert (cp[..., 1] - want.double().mean(-1).float()).abs().max().item() <= 2e-6


@pytest.mark.parametrize("n,c,cs,h,w", [(1, 32, 32, 19, 23), (2, 96, 96, 9, 40), (1, 256, 256, 34, 60), (1, 17, 24, 8, 8)])
def test_cbam_spatial(hip_lib, n, c, cs, h, w):
    """SpatialGate (:469-482): xs *= sigmoid(bn(conv7x7(comp))), in place, channel window respected"""
    import torch.nn.functional as F

    from cfi_amd import _lib

    g = torch.Generator().manual_seed(c + h)
    xs = torch.randn(n, h, w, cs, generator=g)
    comp = torch.randn(n, h, w, 2, generator=g)
    wt = torch.randn(7, 7, 2, generator=g) * 0.1
    bn_a, bn_b = 0.8, -0.1
    s = F.conv2d(comp.permute(0, 3, 1, 2).double(), wt.permute(2, 0, 1)[None].double(), padding=3)[:, 0]      # [n, h, w]
    want = xs.clone()
    want[..., :c] = (xs[..., :c].double() * torch.sigmoid(s * bn_a + bn_b)[..., None]).float()
    xd, cd, wd = xs.cuda(), comp.cuda(), wt.cuda()
    _lib.check(hip_lib.vfi_cbam_spatial(xd.data_ptr(), cs, cd.data_ptr(), wd.data_ptr(), bn_a, bn_b, c, n, h, w, None), "vfi_cbam_spatial")
    torch.cuda.synchronize()
    got = xd.cpu()
    assert (got[..., :c] - want[..., :c]).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item())
    assert torch.equal(got[..., c:], xs[..., c:]), "touched channels outside its window"
