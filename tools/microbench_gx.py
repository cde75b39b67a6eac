"""Grouped 3x3 convolution, 32 channels per group: fp32-MFMA kernels (csrc/gconv32.hip) vs the grouped split-operand MFMA path
(csrc/xconv.hip, csrc/xwgrad3.hip) at the ResNeXt stage-3 / stage-2 shapes of a 16-image chunk."""
import json
import sys

import torch

sys.path.insert(0, 'dynamic-video-depth_amd')
from dvd_hip import conv as C  # noqa: E402
from tools_timeit import timeit  # noqa: E402


def main():
    for (N, Cc, H, W) in ((16, 1024, 24, 42), (16, 512, 48, 84)):
        torch.manual_seed(0)
        G = Cc // 32
        x = torch.randn(N, Cc, H, W, device='cuda').requires_grad_(True)
        w = (torch.randn(Cc, 32, 3, 3, device='cuda') * 0.05).requires_grad_(True)
        gy = torch.randn(N, Cc, H, W, device='cuda')
        flop = 2.0 * N * Cc * 32 * 9 * H * W
        rec = {'shape': [N, Cc, H, W], 'gflop': flop / 1e9}

        def run(fn):
            y = fn()
            fwd = timeit(lambda: fn(), 10)
            tot = timeit(lambda: torch.autograd.grad(fn(), (x, w), gy), 10)
            return fwd, tot - fwd, y

        f1, b1, y1 = run(lambda: C.gconv3x3_c32(x, w))
        f2, b2, y2 = run(lambda: C._xconv(x, w, None, None, False, False, G))
        rec.update(gconv32_fwd_ms=f1, gconv32_bwd_ms=b1, xconv_fwd_ms=f2, xconv_bwd_ms=b2,
                   max_diff=float((y1 - y2).abs().max() / y1.abs().max()))
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
