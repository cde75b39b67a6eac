"""Host logic of the fused gradient joins / ReLU-mask hand-over (dvd_hip/conv.py `_XConvBn`, `_XConv`, `_Site`, `alias`) on
the CPU: the HIP entry points the two autograd Functions call are replaced by plain-torch stand-ins of the same contracts
(convolution + epilogue operands, BatchNorm mask / channel-sum pass, the per-site finalisation), so that what is tested is
the WIRING -- which tensor is handed to which launch as `residual` / `mask_src`, when a site may skip its mask pass, what
every backward returns -- against autograd on the ATen expression of the same block (the ResNeXt bottleneck behind
third_party/midas_blocks.py:35-50, the ResidualConvUnit of midas_blocks.py:102-135).  The kernels themselves are tested on
the GPU (tests/test_06_xconv_gpu.py, tests/test_09_fused_joins_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F


class _FakeLib(object):
    """Stand-ins that receive the tensors themselves (conv._p is patched to the identity)."""

    def dvd_bnrelu_bwd_workspace_bytes(self, N, C, HW):
        return 16

    def dvd_bnrelu_bwd_t(self, gy, y, x, gamma, mean, var, eps, gx, gres, ggamma, gbeta, ws, ws_bytes, f16, out_scale, N, C, HW,
                         relu, g_amax, stream):
        assert not f16 and out_scale is None          # fp32 storage here (the fp16 variants: tests/test_10_act_fp16_gpu.py)
        g = gy * (y > 0).to(gy.dtype) if relu else gy
        if gres is not None:
            gres.copy_(g)
        if gbeta is not None:
            gbeta.copy_(g.sum((0, 2, 3)))
        if g_amax is not None:
            g_amax.fill_(float(g.abs().max()))
        self.mask_passes = getattr(self, 'mask_passes', 0) + int(bool(relu))
        self.sum_passes = getattr(self, 'sum_passes', 0) + int(not relu)
        return 0

    def dvd_convbn_finalize(self, W, dW, dbeta, gamma, mean, var, eps, cbias, Cout, K, dgamma, dcbias, stream):
        rstd = 1.0 / torch.sqrt(var + eps)
        s = (gamma if gamma is not None else torch.ones_like(var)) * rstd
        acc = (W.reshape(Cout, -1) * dW.reshape(Cout, -1)).sum(1)
        dW.mul_(s.reshape(-1, 1, 1, 1))
        if dgamma is not None:
            dgamma.copy_(rstd * (acc + ((cbias if cbias is not None else 0.0) - mean) * dbeta))
        if dcbias is not None:
            dcbias.copy_(s * dbeta)
        return 0


@pytest.fixture
def fake_kernels(monkeypatch):
    from dvd_hip import conv as C
    lib = _FakeLib()
    calls = {'residual': 0, 'mask_src': 0}

    def run(x, packed, Cout, KS, bias=None, residual=None, mask_src=None, relu_in=False, relu_out=False, res_relu=False,
            groups=1, bn=None, x_amax=None, y_amax=None):
        w, transposed = packed
        xin = x.relu() if relu_in else x
        if transposed:
            y = F.conv_transpose2d(xin, w, padding=KS // 2, groups=groups)
        else:
            y = F.conv2d(xin, w, bias, padding=KS // 2, groups=groups)
        if bn is not None:
            g, b, m, v, eps = bn
            s = (g if g is not None else 1.0) / torch.sqrt(v + eps)
            y = y * s.reshape(1, -1, 1, 1) + ((b if b is not None else 0.0) - m * s).reshape(1, -1, 1, 1)
        if residual is not None:                      # the epilogue's order: residual, then mask, then ReLU
            calls['residual'] += 1
            y = y + (residual.relu() if res_relu else residual)
        if mask_src is not None:
            calls['mask_src'] += 1
            y = y * (mask_src > 0).to(y.dtype)
        if relu_out:
            y = y.relu()
        if y_amax is not None:
            y_amax.fill_(float(y.abs().max()))
        return y.contiguous()

    def scaled(weight, groups, gamma, var, eps):
        s = (gamma if gamma is not None else torch.ones_like(var)) / torch.sqrt(var + eps)
        return (weight.detach() * s.reshape(-1, 1, 1, 1), True)

    def wgrad(x, gy, wshape, relu_in, groups=1, x_amax=None, g_amax=None, rowsum=None):
        xin = x.relu() if relu_in else x
        gw = torch.nn.grad.conv2d_weight(xin, wshape, gy, padding=wshape[2] // 2, groups=groups)
        if rowsum is not None:
            rowsum.copy_(gy.sum((0, 2, 3)))
        return gw

    monkeypatch.setattr(C, '_xconv_run', run)
    monkeypatch.setattr(C, 'xconv_packed', lambda weight, transposed, groups=1: (weight.detach(), bool(transposed)))
    monkeypatch.setattr(C, 'xconv_packed_scaled', scaled)
    monkeypatch.setattr(C, 'xconv_wgrad', wgrad)
    monkeypatch.setattr(C, 'amax_of', lambda t: t.detach().abs().max().reshape(1))
    monkeypatch.setattr(C, 'new_scalar', lambda device: torch.zeros(1, dtype=torch.float64))
    monkeypatch.setattr(C, 'set_amax', lambda t, am: t)
    monkeypatch.setattr(C, 'known_amax', lambda t: None)
    monkeypatch.setattr(C, '_p', lambda t: t)
    monkeypatch.setattr(C, '_stream', lambda: 0)
    monkeypatch.setattr(C, '_workspace', lambda nbytes, device: torch.empty(int(nbytes), dtype=torch.uint8))
    monkeypatch.setattr(C._lib, 'load', lambda: lib)
    monkeypatch.setattr(C._lib, 'check', lambda rc, name: None)
    for k in C.STATS:
        C.STATS[k] = 0
    return C, lib, calls


def _bn_params(C_, g):
    return (1.0 + 0.1 * torch.randn(C_, generator=g, dtype=torch.float64), 0.05 * torch.randn(C_, generator=g, dtype=torch.float64),
            0.1 * torch.randn(C_, generator=g, dtype=torch.float64), 0.5 + torch.rand(C_, generator=g, dtype=torch.float64))


def _site(C, x, w, bn, residual=None, relu=True, alias=False, eps=1e-5):
    """What conv.conv_bn_act does on the fused path (its `x.is_cuda` gate is the only thing skipped here)."""
    gamma, beta, mean, var = bn
    in_site = getattr(x, '_dvd_site', None)
    out_site = C._Site() if relu else None
    out = C._XConvBn.apply(x, C.amax_of(x), w, None, gamma, beta, mean, var, eps, residual, relu, 1, alias, in_site, out_site)
    out[0]._dvd_site = out_site
    return (out[0], out[2]) if alias else out[0]


def _ref_site(x, w, bn, residual=None, relu=True, eps=1e-5):
    gamma, beta, mean, var = bn
    y = F.batch_norm(F.conv2d(x, w, padding=w.shape[2] // 2), mean, var, gamma, beta, False, 0.0, eps)
    if residual is not None:
        y = y + residual
    return y.relu() if relu else y


@pytest.mark.parametrize('shortcut_conv', [False, True])
def test_bottleneck_joins_and_masks(fake_kernels, shortcut_conv):
    C, lib, calls = fake_kernels
    g = torch.Generator().manual_seed(3)
    ch, mid = 6, 4
    ws = [torch.randn(mid, ch, 1, 1, generator=g, dtype=torch.float64) * 0.4, torch.randn(mid, mid, 3, 3, generator=g, dtype=torch.float64) * 0.2,
          torch.randn(ch, mid, 1, 1, generator=g, dtype=torch.float64) * 0.4, torch.randn(ch, ch, 1, 1, generator=g, dtype=torch.float64) * 0.4]
    bns = [_bn_params(c, g) for c in (mid, mid, ch, ch)]
    x0 = torch.randn(2, ch, 5, 7, generator=g, dtype=torch.float64)
    gy = torch.randn(2, ch, 5, 7, generator=g, dtype=torch.float64)

    def leaves():
        return ([w.clone().requires_grad_(True) for w in ws],
                [tuple(t.clone().requires_grad_(i < 2) for i, t in enumerate(bn)) for bn in bns], x0.clone().requires_grad_(True))

    # reference: the ATen expression, two blocks in a row (the second block's first convolution is the consumer of the
    # first block's output site)
    W, B, x = leaves()
    h = _ref_site(x, W[3], B[3])                                     # a producer site in front: x of block 1 is a ReLU output
    for _ in range(2):
        y = _ref_site(_ref_site(_ref_site(h, W[0], B[0]), W[1], B[1]), W[2], B[2], relu=False)
        skip = _ref_site(h, W[3], B[3], relu=False) if shortcut_conv else h
        h = (y + skip).relu()
    h.backward(gy)
    want = [x.grad] + [w.grad for w in W] + [t.grad for bn in B for t in bn[:2]]

    W, B, x = leaves()
    h = _site(C, x, W[3], B[3])
    for _ in range(2):
        y, ha = _site(C, h, W[0], B[0], alias=True)                 # MiDaS._Bottleneck.forward
        y = _site(C, y, W[1], B[1])
        skip = _site(C, ha, W[3], B[3], relu=False) if shortcut_conv else ha
        h = _site(C, y, W[2], B[2], residual=skip)
    h.backward(gy)
    got = [x.grad] + [w.grad for w in W] + [t.grad for bn in B for t in bn[:2]]
    for i, (a, b) in enumerate(zip(got, want)):
        # (the Function keeps the channel sums in fp32 whatever the data's dtype: 1e-7-level differences in the BatchNorm
        # gradients; a wiring error is an O(1) difference)
        assert torch.allclose(a, b.to(a.dtype), rtol=1e-5, atol=1e-6), (i, float((a - b).abs().max()))
    # every site but the last (nobody consumes its output here) found its mask applied by its consumer: 1 + 2 * 3 - 1 sites
    assert C.STATS['sites_premasked'] == 6 and C.STATS['sites_masked'] == 1 and lib.mask_passes == 1
    # the joins: each block's first convolution received the shortcut's gradient as its backward-data residual operand
    assert calls['residual'] >= 2 + 2 and calls['mask_src'] == 6      # (forward residuals of the two blocks + two joins)


def test_residual_conv_unit_shares_the_mask(fake_kernels):
    C, lib, calls = fake_kernels
    g = torch.Generator().manual_seed(5)
    w1 = torch.randn(4, 4, 3, 3, generator=g, dtype=torch.float64) * 0.3
    w2 = torch.randn(4, 4, 3, 3, generator=g, dtype=torch.float64) * 0.3
    b1, b2 = torch.randn(4, generator=g, dtype=torch.float64), torch.randn(4, generator=g, dtype=torch.float64)
    x0 = torch.randn(2, 4, 6, 5, generator=g, dtype=torch.float64)
    gy = torch.randn(2, 4, 6, 5, generator=g, dtype=torch.float64)

    def leaves():
        return [t.clone().requires_grad_(True) for t in (x0, w1, b1, w2, b2)]
    x, a1, c1, a2, c2 = leaves()
    h = x * 1.0
    (F.conv2d(F.conv2d(h.relu(), a1, c1, padding=1).relu(), a2, c2, padding=1) + h.relu()).backward(gy)
    want = [t.grad for t in (x, a1, c1, a2, c2)]
    x, a1, c1, a2, c2 = leaves()
    h = x * 1.0
    y, ha = C._xconv(h, a1, c1, None, True, False, 1, alias=True)                               # MiDaS.ResidualConvUnit.forward
    out = C._xconv(y, a2, c2, ha, True, True, 1, res_unmasked=True)
    out.backward(gy)
    for a, b in zip([t.grad for t in (x, a1, c1, a2, c2)], want):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-11)
    assert calls['residual'] == 2 and lib.__dict__.get('mask_passes', 0) == 0       # forward skip + the join; no ATen-style mask pass


def test_encoder_level_feeds_projection_and_next_level(fake_kernels):
    """MidasNet.forward: a level's output (a BatchNorm+ReLU site) feeds its decoder projection (plain convolution) and, through
    the projection's alias, the next level: the projection's backward-data launch adds the next level's gradient and applies
    the site's mask to the sum."""
    C, lib, calls = fake_kernels
    g = torch.Generator().manual_seed(8)
    w0 = torch.randn(5, 3, 1, 1, generator=g, dtype=torch.float64) * 0.5
    wp = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64) * 0.2
    w1 = torch.randn(6, 5, 1, 1, generator=g, dtype=torch.float64) * 0.4
    bn0, bn1 = _bn_params(5, g), _bn_params(6, g)
    x0 = torch.randn(2, 3, 6, 4, generator=g, dtype=torch.float64)
    gp, gn = torch.randn(2, 4, 6, 4, generator=g, dtype=torch.float64), torch.randn(2, 6, 6, 4, generator=g, dtype=torch.float64)

    def leaves():
        return [t.clone().requires_grad_(True) for t in (x0, w0, wp, w1)]
    x, a0, ap, a1 = leaves()
    lvl = _ref_site(x, a0, bn0)
    (F.conv2d(lvl, ap, padding=1) * gp).sum().add((_ref_site(lvl, a1, bn1) * gn).sum()).backward()
    want = [t.grad for t in (x, a0, ap, a1)]
    x, a0, ap, a1 = leaves()
    lvl = _site(C, x, a0, bn0)
    r, lvl_a = C._xconv(lvl, ap, None, None, False, False, 1, alias=True)
    ((r * gp).sum() + (_site(C, lvl_a, a1, bn1) * gn).sum()).backward()
    for i, (a, b) in enumerate(zip([t.grad for t in (x, a0, ap, a1)], want)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (i, float((a - b).abs().max()))
    # the level's site found its mask applied (by the projection's epilogue, on the SUM of the two gradients)
    assert C.STATS['sites_premasked'] == 1 and C.STATS['sites_masked'] == 1 and calls['mask_src'] == 1
