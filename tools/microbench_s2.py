#!/usr/bin/env python
"""The three stride-2 3x3 convolutions of ResNeXt-101 32x8d (Bottleneck.conv2 of layer2.0 / layer3.0 / layer4.0) at the bench's
48-image launches: forward + backward (input and weight gradients) through the module, native strided kernels (csrc/xconv.hip
XArgs::S2 / ZI) against the stride-1 kernels with sub-sampling / zero-interleaving around them (DVD_AB=no_s2)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
from dvd_hip import conv as C  # noqa: E402

SHAPES = [(48, 512, 32, 96, 168), (48, 1024, 32, 48, 84), (48, 2048, 32, 24, 42)]   # N, C, groups, H, W (input)


def timeit(fn, iters=8):
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
    half = bool(os.environ.get('XCONV_FP16'))
    if half:
        from dvd_hip import ops
        C.set_grad_scale_state(ops.gscale_new(torch.device('cuda')))
    for (N, Cc, G, H, W) in SHAPES:
        torch.manual_seed(0)
        mod = (C.GroupedConv3x3C16(Cc, stride=2) if Cc // G == 16 else C.XConv2d(Cc, Cc, 3, stride=2, padding=1, groups=G,
                                                                                    bias=False)).cuda()
        x = torch.randn(N, Cc, H, W, device='cuda')
        gy = torch.randn(N, Cc, (H + 1) // 2, (W + 1) // 2, device='cuda')
        if half:
            x, gy = x.half(), gy.half()
        x.requires_grad_(True)
        rec = {'shape': [N, Cc, G, H, W], 'act': 'fp16' if half else 'fp32'}
        for name, off in (('native', False), ('stride1', True)):
            C.AB['no_s2'] = off
            with torch.no_grad():
                rec[name + '_fwd_ms'] = timeit(lambda: mod(x))

            def both():
                x.grad = None
                mod.weight.grad = None
                mod(x).backward(gy)
            rec[name + '_fwd_bwd_ms'] = timeit(both)
        C.AB['no_s2'] = False
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
