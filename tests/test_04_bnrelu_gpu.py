"""Fused eval-mode BatchNorm (+ residual) (+ ReLU) against the ATen ops it replaces (fp32 CPU):
forward, gradients w.r.t. input, residual, gamma, beta.  Tolerance 2e-6 of max|.| forward / input
gradients (x*s+b versus (x-mean)*invstd*gamma+beta rounding), 2e-5 for the channel sums."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N,C,H,W,res,relu', [(2, 8, 12, 20, False, True), (3, 5, 7, 9, True, True),
                                              (2, 16, 24, 42, True, True), (1, 4, 96, 168, False, True),
                                              (2, 6, 5, 5, False, False),
                                              # small planes: several images of a channel per block in the backward pass
                                              (8, 512, 12, 21, True, True), (6, 700, 7, 9, False, True)])
def test_matches_aten(N, C, H, W, res, relu):
    from dvd_hip.conv import bn_eval_relu
    g = torch.Generator().manual_seed(C * 100 + H)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.3 * torch.randn(C, generator=g))
        bn.bias.copy_(0.2 * torch.randn(C, generator=g))
        bn.running_mean.copy_(0.5 * torch.randn(C, generator=g))
        bn.running_var.copy_(0.5 + torch.rand(C, generator=g))
    x = torch.randn(N, C, H, W, generator=g)
    r = torch.randn(N, C, H, W, generator=g) if res else None
    up = torch.randn(N, C, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    y_ref = bn(xr)
    if res:
        y_ref = y_ref + rr
    if relu:
        y_ref = F.relu(y_ref)
    y_ref.backward(up)
    want = {'y': y_ref.detach(), 'gx': xr.grad, 'gw': bn.weight.grad.clone(), 'gb': bn.bias.grad.clone()}
    if res:
        want['gr'] = rr.grad
    bn.zero_grad()
    bng = torch.nn.BatchNorm2d(C).eval().cuda()
    bng.load_state_dict(bn.state_dict())
    xg = x.cuda().requires_grad_(True)
    rg = r.cuda().requires_grad_(True) if res else None
    y = bn_eval_relu(bng, xg, residual=rg, relu=relu)
    # round 6: the pass reports max|y| (the consuming convolution's operand scale) -- exactly, not a bound
    from dvd_hip.ops import known_amax
    assert known_amax(y) is not None and float(known_amax(y)) == float(y.detach().abs().max())
    seen = {}
    y.register_hook(lambda g: None)
    xg.register_hook(lambda g: seen.__setitem__('gx', (known_amax(g), g.detach().abs().max())))
    if res:
        rg.register_hook(lambda g: seen.__setitem__('gr', (known_amax(g), g.detach().abs().max())))
    y.backward(up.cuda())
    for k, (am, true_max) in seen.items():      # ... and max|gx|, max|g_residual| ride on the gradients it hands on
        assert am is not None and float(am) == float(true_max), k
    assert 'gx' in seen
    got = {'y': y.detach(), 'gx': xg.grad, 'gw': bng.weight.grad, 'gb': bng.bias.grad}
    if res:
        got['gr'] = rg.grad
    for k, tol in (('y', 2e-6), ('gx', 2e-6), ('gr', 2e-6), ('gw', 2e-5), ('gb', 2e-5)):
        if k not in want:
            continue
        a, b = got[k].cpu().numpy(), want[k].numpy()
        assert np.abs(a - b).max() <= tol * np.abs(b).max() + 1e-6, k
    # training-mode BN is not the fused path: it must still behave like ATen's
    bng.train()
    out = bn_eval_relu(bng, x.cuda())
    assert out.shape == x.shape and bool((out >= 0).all())
