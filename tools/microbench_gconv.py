#!/usr/bin/env python
"""Times the grouped-convolution kernels (8 and 32 channels per group) at the MiDaS shapes and at the
doubled resolution of BASELINE config 5; prints ms per call and TFLOP/s next to MIOpen's F.conv2d."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
os.environ.setdefault('MIOPEN_LOG_LEVEL', '1')
from dvd_hip.conv import gconv3x3_c8, gconv3x3_c32  # noqa: E402


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    for cpg, (N, C, H, W) in ((8, (16, 256, 96, 168)), (32, (16, 1024, 24, 42)), (32, (16, 1024, 48, 84)),
                              (32, (16, 512, 48, 84)), (32, (4, 1024, 96, 168))):
        op = gconv3x3_c8 if cpg == 8 else gconv3x3_c32
        x = torch.randn(N, C, H, W, device='cuda', requires_grad=True)
        w = (torch.randn(C, cpg, 3, 3, device='cuda') / 10).requires_grad_(True)
        gy = torch.randn(N, C, H, W, device='cuda')
        flops = 2.0 * 9 * cpg * C * H * W * N

        def fwd():
            with torch.no_grad():
                op(x, w)

        def fwd_bwd():
            y = op(x, w)
            y.backward(gy)
            x.grad = w.grad = None

        def ref_fwd():
            with torch.no_grad():
                F.conv2d(x, w, None, 1, 1, 1, C // cpg)

        t_f, t_fb, t_ref = timeit(fwd), timeit(fwd_bwd), timeit(ref_fwd)
        print('cpg %2d %s: fwd %.3f ms (%.1f TF/s)  fwd+bwd %.3f ms (%.1f TF/s)  MIOpen fwd %.3f ms' % (
            cpg, (N, C, H, W), t_f, flops / t_f / 1e9, t_fb, 3 * flops / t_fb / 1e9, t_ref))


if __name__ == '__main__':
    main()
