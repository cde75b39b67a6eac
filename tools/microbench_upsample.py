"""Bilinear x2 up-sampling kernels at the MiDaS decoder's sizes (16-image chunk at 384x672): GB/s of 4 B written + 1 B read per
output element."""
import json
import sys

import torch

sys.path.insert(0, 'dynamic-video-depth_amd')
sys.path.insert(0, 'tools')
from dvd_hip.conv import upsample_bilinear2x  # noqa: E402
from tools_timeit import timeit  # noqa: E402

for (N, C, H, W, align) in ((16, 256, 96, 168, True), (16, 256, 48, 84, True), (16, 128, 192, 336, False)):
    x = torch.randn(N, C, H, W, device='cuda', requires_grad=True)
    gy = torch.randn(N, C, 2 * H, 2 * W, device='cuda')
    fwd = timeit(lambda: upsample_bilinear2x(x, align), 10)
    tot = timeit(lambda: torch.autograd.grad(upsample_bilinear2x(x, align), x, gy), 10)
    gb = N * C * H * W * 4 * 5 / 1e9
    print(json.dumps({'shape': [N, C, H, W], 'align': align, 'fwd_ms': fwd, 'fwd_GBps': gb / fwd * 1e3, 'bwd_ms': tot - fwd,
                      'bwd_GBps': gb / (tot - fwd) * 1e3}))
