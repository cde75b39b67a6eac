#!/usr/bin/env python
"""The hourglass's large-kernel inception branches (third_party/hourglass.py:21-57: 5x5 / 7x7 / 11x11 over 32 or 64 channels) on
csrc/xconv.hip's direct-staging kernels: forward and backward-data at 16-image launches."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
from dvd_hip import conv as C  # noqa: E402

SHAPES = [(16, 64, 16, 384, 672, 11), (16, 64, 16, 384, 672, 7), (16, 32, 32, 192, 336, 7), (16, 32, 64, 96, 168, 5),
          (16, 64, 64, 48, 84, 11), (16, 32, 32, 192, 336, 5)]      # N, Cin, Cout, H, W, k


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    for (N, Ci, Co, H, W, k) in SHAPES:
        torch.manual_seed(0)
        conv = torch.nn.Conv2d(Ci, Co, k, padding=k // 2, bias=False).cuda()
        x, gy = torch.randn(N, Ci, H, W, device='cuda'), torch.randn(N, Co, H, W, device='cuda')
        pk, pkT = C.xconv_packed(conv.weight, False), C.xconv_packed(conv.weight, True)
        gf = 2.0 * N * Ci * Co * k * k * H * W / 1e9
        rec = {'shape': [N, Ci, Co, H, W, k], 'gflop': gf}
        rec['fwd_ms'] = timeit(lambda: C._xconv_run(x, pk, Co, k))
        rec['dgrad_ms'] = timeit(lambda: C._xconv_run(gy, pkT, Ci, k))
        rec['fwd_tfs'] = gf / rec['fwd_ms']
        rec['dgrad_tfs'] = gf / rec['dgrad_ms']
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
