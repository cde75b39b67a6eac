"""Structural pin of the MiDaS encoder: the product's ResNeXt-101 32x8d
(dvd_hip/third_party/MiDaS.py) against the independent restatement of torchvision 0.10's published architecture in
oracle/resnext.py -- state_dict keys, shapes, parameter count (86 742 336) and a seeded
forward / backward on CPU.  The golden MiDaS full-step fixtures are generated with the ORACLE's encoder patched into
the reference (tests/golden/make_golden.py), not the product's, so the GPU full-step tests are not circular."""
import numpy as np
import torch
from torch import nn

import helpers
from oracle import resnext


def _assemble(enc):
    """What the reference's _make_resnet_backbone does with the hub object (third_party/midas_blocks.py:35-45)."""
    p = nn.Module()
    p.layer1 = nn.Sequential(enc.conv1, enc.bn1, enc.relu, enc.maxpool, enc.layer1)
    p.layer2, p.layer3, p.layer4 = enc.layer2, enc.layer3, enc.layer4
    return p


def test_encoder_state_dict_matches_the_independent_restatement():
    from dvd_hip.third_party.MiDaS import make_resnext101_32x8d_backbone
    ours = make_resnext101_32x8d_backbone()
    ref = _assemble(resnext.resnext101_32x8d())
    a, b = ours.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape), k
    assert sum(p.numel() for p in ref.parameters()) == 86742336
    assert sum(p.numel() for p in ours.parameters()) == 86742336
    # per-module hyper-parameters that the shapes do not pin: stride / padding / groups of every convolution
    ma = {k: m for k, m in ours.named_modules() if isinstance(m, nn.Conv2d)}
    mb = {k: m for k, m in ref.named_modules() if isinstance(m, nn.Conv2d)}
    assert ma.keys() == mb.keys()
    for k in ma:
        for attr in ('stride', 'padding', 'dilation', 'groups', 'kernel_size'):
            assert tuple(np.atleast_1d(getattr(ma[k], attr))) == tuple(np.atleast_1d(getattr(mb[k], attr))), (k, attr)
        assert (ma[k].bias is None) == (mb[k].bias is None), k
    pa = {k: m for k, m in ours.named_modules() if isinstance(m, nn.MaxPool2d)}
    pb = {k: m for k, m in ref.named_modules() if isinstance(m, nn.MaxPool2d)}
    assert pa.keys() == pb.keys() and len(pa) == 1
    for k in pa:
        assert (pa[k].kernel_size, pa[k].stride, pa[k].padding) == (pb[k].kernel_size, pb[k].stride, pb[k].padding)


def test_encoder_forward_backward_matches_on_seeded_weights():
    from dvd_hip.third_party.MiDaS import make_resnext101_32x8d_backbone
    ours = helpers.seeded_fill_(make_resnext101_32x8d_backbone(), 7).eval()
    ref = helpers.seeded_fill_(_assemble(resnext.resnext101_32x8d()), 7).eval()
    torch.manual_seed(3)
    x = torch.rand(1, 3, 64, 96)
    outs = []
    for net in (ours, ref):
        xi = x.clone().requires_grad_(True)
        l1 = net.layer1(xi)
        l2 = net.layer2(l1)
        l3 = net.layer3(l2)
        l4 = net.layer4(l3)
        (l1.mean() + l2.mean() + l3.mean() + l4.mean()).backward()
        outs.append(([t.detach() for t in (l1, l2, l3, l4)], xi.grad,
                     {k: p.grad for k, p in net.named_parameters()}))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert a.shape == b.shape
        assert helpers.rel_err(a.numpy(), b.numpy()) < 1e-6        # same ATen kernels on both sides
    assert helpers.rel_err(outs[0][1].numpy(), outs[1][1].numpy()) < 1e-5
    for k, g in outs[1][2].items():
        assert helpers.rel_err(outs[0][2][k].numpy(), g.numpy()) < 1e-5, k
