"""Grouped 3x3 convolution kernels (8 channels per group) against torch's fp32 CPU
convolution: forward, backward-data, backward-weight.  Tolerance: rtol 1e-5 of max|.|
forward / backward-data (72-term fp32 dot products in a different order), 2e-5 for the weight
gradient (sums over N*H*W pixels)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, gy):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1, 1, x.shape[1] // 8)
    y.backward(gy)
    return y.detach(), x.grad, w.grad


@pytest.mark.parametrize('N,C,H,W', [(2, 16, 20, 36), (1, 256, 48, 84), (3, 8, 17, 67), (2, 32, 16, 64),
                                     (1, 24, 5, 3), (2, 256, 96, 168)])
def test_matches_torch_cpu(N, C, H, W):
    from dvd_hip.conv import GroupedConv3x3C8, gconv3x3_c8
    g = torch.Generator().manual_seed(N * 1000 + C + H + W)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 8, 3, 3, generator=g) / 8.0
    gy = torch.randn(N, C, H, W, generator=g)
    y_ref, gx_ref, gw_ref = _ref(x, w, gy)
    xg = x.cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    y = gconv3x3_c8(xg, wg)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    for name, got, want, tol in (('y', y, y_ref, 1e-5), ('gx', xg.grad, gx_ref, 1e-5), ('gw', wg.grad, gw_ref, 2e-5)):
        got = got.detach().cpu().numpy()
        want = want.numpy()
        assert np.abs(got - want).max() <= tol * np.abs(want).max() + 1e-6, name
    # module form: same parameter name/shape as the nn.Conv2d it replaces, deterministic weight gradient
    m = GroupedConv3x3C8(C).cuda()
    assert tuple(m.weight.shape) == (C, 8, 3, 3) and list(m.state_dict()) == ['weight']
    with torch.no_grad():
        m.weight.copy_(w)
    a = m(x.cuda())
    a.backward(gy.cuda())
    g1 = m.weight.grad.clone()
    m.weight.grad = None
    m(x.cuda()).backward(gy.cuda())
    assert torch.equal(g1, m.weight.grad)
    assert torch.equal(a, y.detach())


def test_midas_stage1_uses_the_hip_kernels():
    from dvd_hip.conv import GroupedConv3x3C8
    from dvd_hip.third_party.MiDaS import MidasNet
    net = MidasNet()
    blocks = list(net.pretrained.layer1[4])
    assert len(blocks) == 3 and all(isinstance(b.conv2, GroupedConv3x3C8) for b in blocks)
    assert not isinstance(net.pretrained.layer2[0].conv2, GroupedConv3x3C8)     # stride 2: MIOpen
    assert 'pretrained.layer1.4.0.conv2.weight' in net.state_dict()


@pytest.mark.parametrize('N,C,H,W', [(2, 32, 9, 13), (1, 64, 24, 42), (3, 32, 5, 3), (2, 1024, 24, 42), (1, 32, 33, 40),
                                     (1, 32, 20, 84)])
def test_c32_matches_torch_cpu(N, C, H, W):
    """32 channels per group (fp32 MFMA kernels): forward, backward-data, backward-weight against
    torch's fp32 CPU convolution.  288-term dot products per output (fp32, different summation
    order): 2e-5 of max|.|; the weight gradient sums N*H*W products: 5e-5."""
    from dvd_hip.conv import GroupedConv3x3C32, gconv3x3_c32
    g = torch.Generator().manual_seed(N * 1000 + C + H + W)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 32, 3, 3, generator=g) / 17.0
    gy = torch.randn(N, C, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1, 1, C // 32)
    yr.backward(gy)
    xg = x.cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    y = gconv3x3_c32(xg, wg)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    for name, got, want, tol in (('y', y, yr, 2e-5), ('gx', xg.grad, xr.grad, 2e-5), ('gw', wg.grad, wr.grad, 5e-5)):
        got = got.detach().cpu().numpy()
        want = want.detach().numpy()
        assert np.abs(got - want).max() <= tol * np.abs(want).max() + 1e-6, name
    m = GroupedConv3x3C32(C).cuda()
    assert tuple(m.weight.shape) == (C, 32, 3, 3) and list(m.state_dict()) == ['weight']
    with torch.no_grad():
        m.weight.copy_(w)
    m(x.cuda()).backward(gy.cuda())
    g1 = m.weight.grad.clone()
    m.weight.grad = None
    m(x.cuda()).backward(gy.cuda())
    assert torch.equal(g1, m.weight.grad)                      # deterministic weight gradient


def test_midas_stage3_uses_the_mfma_kernels():
    from dvd_hip.conv import GroupedConv3x3C32
    from dvd_hip.third_party.MiDaS import MidasNet
    net = MidasNet()
    blocks = list(net.pretrained.layer3)
    assert len(blocks) == 23 and not isinstance(blocks[0].conv2, GroupedConv3x3C32)     # stride 2: MIOpen
    assert all(isinstance(b.conv2, GroupedConv3x3C32) for b in blocks[1:])
    assert 'pretrained.layer3.5.conv2.weight' in net.state_dict()


def test_c16_pairs_groups_on_the_mfma_kernels():
    """16 channels per group (ResNeXt stage 2) run as block-diagonal 32-channel groups: same values and
    gradients as torch's grouped convolution, same parameter shape."""
    from dvd_hip.conv import GroupedConv3x3C16
    N, C, H, W = 2, 64, 12, 21
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g)
    gy = torch.randn(N, C, H, W, generator=g)
    m = GroupedConv3x3C16(C)
    assert tuple(m.weight.shape) == (C, 16, 3, 3)
    xr = x.clone().requires_grad_(True)
    yr = m(xr)                                    # CPU: F.conv2d with groups = C // 16
    yr.backward(gy)
    want_gw = m.weight.grad.clone()
    m.weight.grad = None
    mg = m.cuda()
    xg = x.cuda().requires_grad_(True)
    y = mg(xg)
    y.backward(gy.cuda())
    for name, got, want, tol in (('y', y, yr, 2e-5), ('gx', xg.grad, xr.grad, 2e-5), ('gw', mg.weight.grad, want_gw, 5e-5)):
        a, b = got.detach().cpu().numpy(), want.detach().numpy()
        assert np.abs(a - b).max() <= tol * np.abs(b).max() + 1e-6, name
