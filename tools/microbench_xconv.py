#!/usr/bin/env python
"""Rates of the dense-convolution kernels (csrc/xconv.hip, csrc/xwgrad.hip) at the MiDaS shapes of a 16-image
chunk at 384x672, next to MIOpen's (F.conv2d) on the same tensors.  TF/s = direct-convolution fp32 FLOPs / time."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
os.environ.setdefault('MIOPEN_LOG_LEVEL', '1')
from dvd_hip import conv as C  # noqa: E402

SHAPES = [  # N, Cin, Cout, H, W, KS
    (16, 2048, 256, 12, 21, 3), (16, 1024, 256, 24, 42, 3), (16, 512, 256, 48, 84, 3), (16, 128, 32, 384, 672, 3),
    (16, 256, 256, 96, 168, 3), (16, 256, 256, 48, 84, 3), (16, 256, 128, 192, 336, 3), (8, 128, 32, 384, 672, 3),
    (16, 256, 256, 96, 168, 1), (16, 1024, 256, 24, 42, 1), (16, 256, 1024, 24, 42, 1), (16, 512, 2048, 12, 21, 1),
]


def timeit(fn, iters):
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
    only = sys.argv[1:]
    shapes = SHAPES[:int(os.environ.get('XCONV_NSHAPES', len(SHAPES)))]
    if os.environ.get('XCONV_ONLY'):                          # comma-separated indices into SHAPES
        shapes = [SHAPES[int(i)] for i in os.environ['XCONV_ONLY'].split(',')]
    nmul = int(os.environ.get('XCONV_NMUL', '1'))            # 3: the 48-image chunks of the bench
    cfg = int(os.environ.get('XCONV_CFG', '0'))              # dvd_xconv_select: 0 auto, 1 round-2 blocks, 2/3 forced, 4 generic loop
    from dvd_hip import _lib
    _lib.check(_lib.load().dvd_xconv_select(cfg), 'dvd_xconv_select')
    skip_wgrad = bool(os.environ.get('XCONV_NO_WGRAD'))
    _lib.check(_lib.load().dvd_xwgrad_select(int(os.environ.get('XWGRAD_VARIANT', '0'))), 'dvd_xwgrad_select')   # 2: round-3 row step
    half = bool(os.environ.get('XCONV_FP16'))                 # fp16 activation storage (configs[4] kernels)
    if half:
        from dvd_hip import ops
        C.set_grad_scale_state(ops.gscale_new(torch.device('cuda')))
    for (N, Cin, Cout, H, W, KS) in shapes:
        N *= nmul
        torch.manual_seed(0)
        x = torch.randn(N, Cin, H, W, device='cuda')
        if os.environ.get('XCONV_ZERO_INPUT'):                # DVFS probe: same code, no data switching
            x.zero_()
        conv = torch.nn.Conv2d(Cin, Cout, KS, padding=KS // 2).cuda()
        gy = torch.randn(N, Cout, H, W, device='cuda')
        if os.environ.get('XCONV_ZERO_INPUT'):
            gy.zero_()
        if half:
            x, gy = x.half(), gy.half()
        flop = 2.0 * N * Cin * Cout * H * W * KS * KS
        rec = {'shape': [N, Cin, Cout, H, W, KS], 'gflop': flop / 1e9, 'act': 'fp16' if half else 'fp32'}
        pk, pkT = C.xconv_packed(conv.weight, False), C.xconv_packed(conv.weight, True)
        it = 10
        rec['xconv_fwd_ms'] = timeit(lambda: C._xconv_run(x, pk, Cout, KS, bias=conv.bias), it)
        rec['xconv_dgrad_ms'] = timeit(lambda: C._xconv_run(gy, pkT, Cin, KS), it)
        if not skip_wgrad:
            rec['xconv_wgrad_ms'] = timeit(lambda: C.xconv_wgrad(x, gy, conv.weight.shape, False), it)
        rec['cfg'] = cfg
        rec['pack_ms'] = timeit(lambda: (conv.weight._dvd_xpack.clear(), C.xconv_packed(conv.weight, False)), it)
        if 'nomiopen' not in only and not half:
            with torch.no_grad():
                rec['miopen_fwd_ms'] = timeit(lambda: F.conv2d(x, conv.weight, conv.bias, padding=KS // 2), it)
            rec['miopen_dgrad_ms'] = timeit(lambda: torch.ops.aten.convolution_backward(
                gy, x, conv.weight, None, [1, 1], [KS // 2] * 2, [1, 1], False, [0, 0], 1, [True, False, False]), it)
            rec['miopen_wgrad_ms'] = timeit(lambda: torch.ops.aten.convolution_backward(
                gy, x, conv.weight, None, [1, 1], [KS // 2] * 2, [1, 1], False, [0, 0], 1, [False, True, False]), it)
        for k in list(rec):
            if k.endswith('_ms') and k != 'pack_ms':
                rec[k.replace('_ms', '_tfs')] = flop / rec[k] / 1e9
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
