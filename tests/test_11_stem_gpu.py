"""The ResNeXt stem on this package's kernels (round 4): `Conv2d(3, 64, 7, 2, 3) -> BatchNorm2d -> ReLU -> MaxPool2d(3, 2, 1)`
(torchvision's ResNet stem behind third_party/midas_blocks.py:35-45) as a 5x5 space-to-depth convolution with the BatchNorm in
its epilogue (dvd_hip.conv.stem_conv_bn_relu, csrc/xconv.hip, csrc/xwgrad.hip KS = 5) and csrc/pool.hip -- against the ATen
modules in float64 on the CPU.  Tolerances: the convolution bounds of tests/test_06_xconv_gpu.py (4e-6 of max forward, 2e-5 of
max for the weight gradient); the max-pool routes gradients exactly like ATen (first maximum of a window)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import log_measured, seeded_fill_

pytestmark = pytest.mark.gpu


def test_space_to_depth_weight_is_the_same_convolution():
    from dvd_hip import conv as C
    torch.manual_seed(0)
    w = torch.randn(8, 3, 7, 7, dtype=torch.float64)
    for (H, W) in ((16, 24), (15, 21)):
        x = torch.randn(2, 3, H, W, dtype=torch.float64)
        want = F.conv2d(x, w, stride=2, padding=3)
        xp = F.pad(x, (0, W % 2, 0, H % 2))
        got = F.conv2d(F.pixel_unshuffle(xp, 2), C.s2d_weight(w), padding=2)
        assert got.shape == want.shape and float((got - want).abs().max()) < 1e-12


@pytest.mark.parametrize('N,H,W', [(2, 64, 96), (1, 48, 84), (2, 33, 51)])
def test_stem_forward_and_gradients(N, H, W):
    from dvd_hip import conv as C
    torch.manual_seed(H)
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
    bn = seeded_fill_(torch.nn.BatchNorm2d(64), 5).eval()
    pool = torch.nn.MaxPool2d(3, 2, 1)
    x = torch.rand(N, 3, H, W)
    cd, bd = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).double(), torch.nn.BatchNorm2d(64).double().eval()
    cd.load_state_dict(conv.state_dict())
    bd.load_state_dict(bn.state_dict())
    yd = pool(F.relu(bd(cd(x.double()))))
    g = torch.randn(yd.shape)
    yd.backward(g.double())
    conv, bn = conv.cuda(), bn.cuda()
    y = C.maxpool3s2(C.stem_conv_bn_relu(conv, bn, x.cuda()))
    assert y.shape == yd.shape
    e = float((y.double().cpu() - yd).abs().max() / yd.abs().max())
    log_measured('stem forward %dx%d' % (H, W), e, 4e-6)
    assert e < 4e-6
    y.backward(g.cuda())
    for name, got, want, tol in (('conv1.weight', conv.weight.grad, cd.weight.grad, 2e-5), ('bn1.weight', bn.weight.grad, bd.weight.grad, 2e-5),
                                 ('bn1.bias', bn.bias.grad, bd.bias.grad, 2e-5)):
        e = float((got.double().cpu() - want).abs().max() / want.abs().max())
        log_measured('stem %s grad %dx%d' % (name, H, W), e, tol)
        assert e < tol, (name, e)


def test_maxpool_routes_gradients_like_aten_including_ties():
    from dvd_hip import conv as C
    torch.manual_seed(3)
    for (N, Cc, H, W) in ((2, 4, 12, 18), (1, 3, 9, 7), (2, 2, 10, 9), (1, 2, 7, 12), (1, 1, 1, 1), (1, 1, 2, 3)):
        x = torch.relu(torch.randn(N, Cc, H, W))          # whole windows of zeros: the tie rule decides
        x[0, 0, :4, :6] = 1.0                              # a plateau of equal maxima
        g = torch.randn(N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool2d(xr, 3, 2, 1)
        yr.backward(g)
        xg = x.cuda().requires_grad_(True)
        y = C.maxpool3s2(xg)
        y.backward(g.cuda())
        assert torch.equal(y.detach().cpu(), yr.detach()) and torch.equal(xg.grad.cpu(), xr.grad)
        y16 = C.maxpool3s2(x.cuda(), to_half=True)
        assert y16.dtype == torch.float16 and torch.equal(y16.cpu(), yr.detach().half())


def test_midas_forward_runs_no_aten_convolution_or_pooling():
    """The depth net's forward + backward launch no MIOpen / ATen convolution, pooling or batch-norm kernel any more."""
    from torch.profiler import ProfilerActivity, profile
    from dvd_hip.third_party.MiDaS import MidasNet, calibrate_head_for_random_init
    torch.manual_seed(0)
    net = calibrate_head_for_random_init(MidasNet(non_negative=True, normalize_input=True)).cuda().eval()
    x = torch.rand(1, 3, 64, 96, device='cuda')
    net(x).sum().backward()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        net(x).sum().backward()
    names = {e.key for e in prof.key_averages()}
    banned = [n for n in names if any(b in n for b in ('aten::convolution', 'aten::miopen', 'aten::max_pool', 'aten::batch_norm',
                                                       'aten::native_batch_norm', 'aten::cudnn'))]
    assert not banned, banned
