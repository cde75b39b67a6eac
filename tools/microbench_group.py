#!/usr/bin/env python
"""Grouped 3x3 convolutions of ResNeXt-101 32x8d stages 2-4 at the bench's 48-image launches: forward, backward-data and
weight gradient through dvd_hip.conv (XCONV_CFG=8: the xconv_kernel path instead of xgroup_kernel)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
from dvd_hip import _lib, conv as C  # noqa: E402

SHAPES = [(48, 512, 32, 48, 84), (48, 1024, 32, 24, 42), (48, 2048, 32, 12, 21)]   # N, C, groups, H, W


def timeit(fn, iters=10):
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
    _lib.check(_lib.load().dvd_xconv_select(int(os.environ.get('XCONV_CFG', '0'))), 'dvd_xconv_select')
    _lib.check(_lib.load().dvd_xwgrad_select(int(os.environ.get('XWGRAD_VARIANT', '0'))), 'dvd_xwgrad_select')
    half = bool(os.environ.get('XCONV_FP16'))
    if half:
        from dvd_hip import ops
        C.set_grad_scale_state(ops.gscale_new(torch.device('cuda')))
    for (N, Cc, G, H, W) in SHAPES:
        torch.manual_seed(0)
        conv = torch.nn.Conv2d(Cc, Cc, 3, padding=1, groups=G, bias=False).cuda()
        x, gy = torch.randn(N, Cc, H, W, device='cuda'), torch.randn(N, Cc, H, W, device='cuda')
        if half:
            x, gy = x.half(), gy.half()
        pk, pkT = C.xconv_packed(conv.weight, False, groups=G), C.xconv_packed(conv.weight, True, groups=G)
        rec = {'shape': [N, Cc, G, H, W], 'act': 'fp16' if half else 'fp32', 'gflop': 2.0 * N * Cc * (Cc // G) * 9 * H * W / 1e9}
        rec['fwd_ms'] = timeit(lambda: C._xconv_run(x, pk, Cc, 3, groups=G))
        rec['dgrad_ms'] = timeit(lambda: C._xconv_run(gy, pkT, Cc, 3, groups=G))
        rec['wgrad_ms'] = timeit(lambda: C.xconv_wgrad(x, gy, conv.weight.shape, False, groups=G))
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
