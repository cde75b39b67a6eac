"""Gradient joins of residual blocks and the ReLU masks of BatchNorm+ReLU sites fused into the backward-data epilogue of
the consuming convolution (dvd_hip/conv.py `_XConv` / `_XConvBn`: `alias`, `_Site`):
the blocks of the MiDaS depth net (third_party/MiDaS.py:164-246 + torchvision's Bottleneck behind midas_blocks.py:35-50,
ResidualConvUnit midas_blocks.py:102-135) on the GPU with both fusions, with autograd's own accumulation and the sites'
own mask passes (`no_alias`, `no_maskfuse`), and on the CPU in float64 through the ATen ops the reference uses.

Tolerances: fused vs unfused are the same arithmetic (the epilogue adds the other consumers' gradient to the exactly
unscaled accumulator: one rounding, like the ATen add) -> 1e-6 of max|.|; against float64 the convolution bounds of
tests/test_06_xconv_gpu.py compounded over the three convolutions of a block: 2e-5 of max|.| (weight gradients 1e-4)."""
import copy

import numpy as np
import pytest
import torch

from helpers import seeded_fill_

pytestmark = pytest.mark.gpu


def _grads(module, x, gy, device, dtype):
    m = copy.deepcopy(module).to(device=device, dtype=dtype).eval()
    xx = x.to(device=device, dtype=dtype).requires_grad_(True)
    # the block input has an upstream producer in the net: give it one here, so that autograd really has two gradients to
    # join for it
    h = xx * 1.0
    y = m(h)
    y.backward(gy.to(device=device, dtype=dtype))
    out = {'y': y.detach().double().cpu(), 'gx': xx.grad.double().cpu()}
    for k, p in m.named_parameters():
        out['g_' + k] = p.grad.double().cpu()
    return out


def _rel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _compare(module, x, gy):
    from dvd_hip import conv as C
    want = _grads(module, x, gy, 'cpu', torch.float64)
    for k in C.STATS:
        C.STATS[k] = 0
    fused = _grads(module, x, gy, 'cuda', torch.float32)
    taken = dict(C.STATS)
    C.AB['no_alias'] = C.AB['no_maskfuse'] = True
    try:
        plain = _grads(module, x, gy, 'cuda', torch.float32)
        C.AB['no_alias'] = False                          # joins fused, ReLU masks in the sites' own passes
        nomask = _grads(module, x, gy, 'cuda', torch.float32)
        C.AB['no_maskfuse'] = False
        C.AB['rowsum'] = True                             # opt-in: channel sums from dvd_xwgrad1s_rowsum, no pass of the site
        rowsum = _grads(module, x, gy, 'cuda', torch.float32)
    finally:
        C.AB['no_alias'] = C.AB['no_maskfuse'] = C.AB['rowsum'] = False
    worst = {}
    for k in want:
        e64, eab = _rel(fused[k], want[k]), max(_rel(fused[k], plain[k]), _rel(fused[k], nomask[k]), _rel(fused[k], rowsum[k]))
        worst[k] = (e64, eab)
        tol = 1e-4 if (k.startswith('g_') and k.endswith('weight') and want[k].dim() == 4) else 2e-5
        assert e64 < tol, '%s: %.2e of max against float64' % (k, e64)
        # the fused joins / masks are the SAME roundings in another launch (epilogue add = ATen add, mask = mask): bit-identical;
        # only the opt-in row-sum variant (another summation order for the per-channel sums) is compared within 1e-6 of max
        assert torch.equal(fused[k], plain[k]) and torch.equal(fused[k], nomask[k]), \
            '%s: fused and unfused joins differ (%.2e / %.2e of max)' % (k, _rel(fused[k], plain[k]), _rel(fused[k], nomask[k]))
        assert eab < 1e-6, '%s: the row-sum variant differs by %.2e of max' % (k, eab)
    print('worst vs float64 %.2e, fused vs unfused %.2e, sites %s' % (max(v[0] for v in worst.values()),
                                                                    max(v[1] for v in worst.values()), taken))
    return taken


@pytest.mark.parametrize('c_in,planes,stride,down', [(256, 64, 1, False),     # stage 1, identity shortcut (8 per group)
                                                     (64, 64, 1, True),       # stage 1 entry: 1x1 shortcut convolution
                                                     (1024, 256, 1, False),   # stage 3 (32 per group: grouped xconv)
                                                     (256, 128, 2, True)])    # stage 2 entry: stride 2, strided shortcut
def test_resnext_bottleneck(c_in, planes, stride, down):
    from dvd_hip.third_party.MiDaS import _Bottleneck
    blk = seeded_fill_(_Bottleneck(c_in, planes, stride, 32, 8, down), 3)
    g = torch.Generator().manual_seed(c_in + planes)
    x = torch.randn(2, c_in, 12, 20, generator=g).relu()          # a block's input is a ReLU output
    gy = torch.randn(2, planes * 4, 12 // stride, 20 // stride, generator=g)
    taken = _compare(blk, x, gy)
    if planes == 256:      # stage 3: both inner sites feed convolutions of the xconv family -> their masks come pre-applied;
        assert taken == {'sites_no_pass': 0, 'sites_premasked': 2, 'sites_masked': 1}, taken


def test_residual_conv_unit_and_fusion_block():
    from dvd_hip.third_party.MiDaS import FeatureFusionBlock, ResidualConvUnit
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 11, 18, generator=g)
    _compare(seeded_fill_(ResidualConvUnit(64), 4), x, torch.randn(2, 64, 11, 18, generator=g))

    class Two(torch.nn.Module):          # FeatureFusionBlock with both inputs derived from one tensor
        def __init__(self):
            super().__init__()
            self.f = FeatureFusionBlock(64)

        def forward(self, t):
            return self.f(t * 0.5, t + 1.0)
    _compare(seeded_fill_(Two(), 6), x, torch.randn(2, 64, 22, 36, generator=g))


def test_hourglass_inception_branches():
    """Four branches read one input (third_party/hourglass.py:21-57): three joins chained through the branches' aliases."""
    from dvd_hip.third_party.hourglass import A2, inception
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 10, 14, generator=g)
    _compare(seeded_fill_(inception(256, A2), 7), x, torch.randn(2, 256, 10, 14, generator=g))
