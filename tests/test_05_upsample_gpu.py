"""Bilinear x2 up-sampling kernels (both align_corners flavours of the MiDaS decoder) against
torch's fp32 CPU F.interpolate: forward and the gradient w.r.t. the input.  Tolerance 2e-6 of
max|.| (same source-index and weight formulas; ATen may contract the blend into FMAs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('align', [True, False])
@pytest.mark.parametrize('N,C,H,W', [(2, 3, 12, 21), (1, 8, 24, 42), (2, 4, 5, 7), (1, 2, 1, 9), (1, 16, 48, 84),
                                     (1, 2, 70, 150), (3, 1, 6, 131),
                                     # separable backward: row segments of 16 that do not divide H, narrow maps
                                     (1, 2, 35, 42), (2, 3, 17, 6), (1, 1, 33, 84), (2, 2, 16, 4)])
def test_matches_torch_cpu(N, C, H, W, align):
    from dvd_hip.conv import upsample_bilinear2x
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(N, C, H, W, generator=g)
    up = torch.randn(N, C, 2 * H, 2 * W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=align)
    yr.backward(up)
    xg = x.cuda().requires_grad_(True)
    y = upsample_bilinear2x(xg, align)
    y.backward(up.cuda())
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(yr.shape)
    for name, got, want in (('y', y.detach(), yr.detach()), ('gx', xg.grad, xr.grad)):
        got, want = got.cpu().numpy(), want.numpy()
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() + 1e-7, name
    # deterministic backward (fixed summation order, no atomics)
    xg2 = x.cuda().requires_grad_(True)
    upsample_bilinear2x(xg2, align).backward(up.cuda())
    assert torch.equal(xg.grad, xg2.grad)


def test_midas_decoder_uses_the_kernels_and_matches_cpu():
    from dvd_hip.third_party.MiDaS import FeatureFusionBlock, Interpolate
    torch.manual_seed(0)
    blk = FeatureFusionBlock(8)
    x = torch.randn(2, 8, 6, 10)
    want = blk(x.clone())
    got = blk.cuda()(x.clone().cuda())
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-5)
    it = Interpolate(2, 'bilinear')
    np.testing.assert_allclose(it(x.cuda()).cpu().numpy(), it(x).numpy(), rtol=1e-5, atol=1e-6)
