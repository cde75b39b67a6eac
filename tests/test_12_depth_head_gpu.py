"""The tail of the MiDaS depth head as HIP passes (round 6): `10000 / clamp(relu(v), min=1e-2)` (csrc/elementwise.hip
dvd_depth_tail_*) and, in front of it, `Conv2d(32, 1, 1)(relu(x))` on the boundary kernel with fp32 features (csrc/a16.hip
dvd_head1x1_*: until round 6 only the fp16-activation mode took that route) -- against the ATen expressions of the reference
(third_party/MiDaS.py:186-195,240-242).  Forward of the tail: bit-exact (torch's reciprocal-then-multiply); backward: 1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_depth_tail_matches_aten_forward_bitwise_and_backward():
    from dvd_hip import conv as C
    g = torch.Generator().manual_seed(3)
    v = (torch.randn(3, 1, 37, 53, generator=g) * 0.05).cuda()      # values on both sides of 0 and of the clamp's 1e-2
    v.view(-1)[:8] = torch.tensor([0.0, 1e-2, 9.999e-3, 1.0001e-2, -1.0, 5.0, 1e-30, -0.0])
    for odd in (0, 3):                                               # an element count that is not a multiple of 4 as well
        a = v.view(-1)[odd:].clone().requires_grad_(True)
        b = a.detach().clone().requires_grad_(True)
        out = C.depth_tail(a)
        ref = 10000 / torch.clamp(F.relu(b), min=1e-2)
        assert torch.equal(out, ref)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).cuda()
        out.backward(go)
        ref.backward(go)
        scale = float(b.grad.abs().max())
        assert float((a.grad - b.grad).abs().max()) <= 1e-6 * scale
        assert torch.equal(a.grad == 0, b.grad == 0)                 # the clamp's mask, element for element


def test_midas_fp32_head_on_the_boundary_kernels_matches_the_sequential_head():
    """The fp32 MiDaS head through conv.head1x1 + conv.depth_tail against the same modules run as the reference's
    nn.Sequential (ReLU / clamp / division on ATen, the one-row 1x1 convolution on the MFMA kernel): depth within 2e-5 (the
    MFMA form splits its operands into two fp16 terms, the boundary kernel is plain fp32 FMAs: closer to the reference), the
    gradient of the head's input within 2e-5 of its largest element."""
    from dvd_hip.third_party.MiDaS import MidasNet, calibrate_head_for_random_init
    torch.manual_seed(0)
    net = calibrate_head_for_random_init(MidasNet(non_negative=True, normalize_input=True)).cuda().eval()
    oc = net.scratch.output_conv
    y0 = torch.randn(2, 256, 24, 40, device='cuda')
    res = []
    for fused in (True, False):
        y = y0.clone().requires_grad_(True)
        for p in oc.parameters():
            p.grad = None
        if fused:
            from dvd_hip import conv as C
            out = C.depth_tail(C.head1x1(oc[4], oc[2](oc[1](oc[0](y))), relu_in=True))
        else:
            out = 10000 / torch.clamp(oc(y), min=1e-2)
        out.backward(torch.ones_like(out) * 1e-3)
        res.append((out.detach(), y.grad.detach(), oc[4].weight.grad.detach().clone(), oc[4].bias.grad.detach().clone()))
    (o1, g1, w1, b1), (o2, g2, w2, b2) = res
    assert float(((o1 - o2).abs() / o2.abs()).max()) < 2e-5
    assert float((g1 - g2).abs().max()) <= 2e-5 * float(g2.abs().max())
    np.testing.assert_allclose(w1.cpu().numpy(), w2.cpu().numpy(), rtol=2e-4, atol=1e-6 * float(w2.abs().max()))
    np.testing.assert_allclose(b1.cpu().numpy(), b2.cpu().numpy(), rtol=2e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('shape', [(2, 5, 12, 20), (1, 3, 7, 9), (3, 2, 1, 1), (2, 4, 24, 42)])
def test_subsample2_equals_strided_slicing_forward_and_backward(shape, dtype):
    """conv._subsample(x, 2) on csrc/pool.hip (dvd_subsample2_*: the stride-2 entries of the ResNeXt stages) against
    x[:, :, ::2, ::2].contiguous() and its autograd backward: both are copies, so bit for bit, odd sizes included."""
    from dvd_hip import conv as C
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g).to(dtype).cuda()
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = C._subsample(a, 2)
    yb = b[:, :, ::2, ::2].contiguous()
    assert ya.is_contiguous() and torch.equal(ya, yb)
    go = torch.randn(yb.shape, generator=g).to(dtype).cuda()
    ya.backward(go)
    yb.backward(go)
    assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('k,st,pad', [(2, 2, 0), (3, 2, 1)])
@pytest.mark.parametrize('shape', [(2, 5, 12, 20), (1, 3, 7, 9), (3, 2, 2, 2), (2, 4, 24, 43), (1, 2, 33, 64)])
def test_avgpool_equals_aten_forward_and_backward(shape, k, st, pad, dtype):
    """conv.AvgPool2d on csrc/pool.hip (dvd_avgpool_*) against nn.AvgPool2d on the same GPU tensors: the hourglass's
    AvgPool2d(2) (third_party/hourglass.py:60-158) and FCNUnet's AvgPool2d(3, 2, 1) (networks/FCNUnet.py:64), odd sizes (floor mode
    drops the last row / column of a 2x2 pool) included.  ATen's arithmetic is reproduced: fp32 bit for bit in the forward
    (same summation order, one division), one fp32 ulp in the backward of overlapping windows (ATen's order over the windows
    is its own); fp16 storage: the fp32 result rounded once."""
    from dvd_hip import conv as C
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g).to(dtype).cuda()
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = C.AvgPool2d(k, st, pad)(a)
    assert type(ya.grad_fn).__name__.startswith('_AvgPool')
    yb = torch.nn.AvgPool2d(k, st, pad)(b)
    assert ya.shape == yb.shape and ya.dtype == yb.dtype
    tol = 1e-3 if dtype == torch.float16 else 2e-7
    if dtype == torch.float32:
        assert torch.equal(ya, yb)
    else:
        assert float((ya.float() - yb.float()).abs().max()) <= tol * float(yb.float().abs().max())
    go = torch.randn(yb.shape, generator=g).to(dtype).cuda()
    ya.backward(go)
    yb.backward(go)
    assert float((a.grad.float() - b.grad.float()).abs().max()) <= tol * float(b.grad.float().abs().max())
    # against float64 on the CPU (the definition)
    xd = x.double().cpu().requires_grad_(True)
    yd = torch.nn.functional.avg_pool2d(xd, k, st, pad)
    yd.backward(go.double().cpu())
    assert float((ya.detach().double().cpu() - yd.detach()).abs().max()) <= (2e-3 if dtype == torch.float16 else 1e-6) * float(yd.abs().max())
    assert float((a.grad.double().cpu() - xd.grad).abs().max()) <= (2e-3 if dtype == torch.float16 else 1e-6) * float(xd.grad.abs().max())
