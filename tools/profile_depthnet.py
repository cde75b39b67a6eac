#!/usr/bin/env python
"""Where does the MiDaS depth net spend its time on MI355X?  One forward + backward of a chunk of
images at the BASELINE resolution under torch.profiler, aggregated per (op, input shapes): which
convolution shapes a hand-written MFMA kernel has to beat, and by how much (FLOP rate per shape)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
os.environ.setdefault('MIOPEN_LOG_LEVEL', '1')


def conv_flops(shapes):
    try:
        x, w = shapes[0], shapes[1]
        n, cin, h, wd = x
        cout, cin_g, kh, kw = w
        return None if not x or not w else (n, cin, h, wd, cout, cin_g, kh, kw)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=8)
    ap.add_argument('--H', type=int, default=384)
    ap.add_argument('--W', type=int, default=672)
    ap.add_argument('--out', default='gpurun_out/depthnet_profile.txt')
    a = ap.parse_args()
    from dvd_hip.third_party.MiDaS import MidasNet, calibrate_head_for_random_init
    torch.manual_seed(0)
    net = calibrate_head_for_random_init(MidasNet(non_negative=True, normalize_input=True)).cuda().eval()
    x = torch.rand(a.images, 3, a.H, a.W, device='cuda')
    g = torch.randn(a.images, 1, a.H, a.W, device='cuda')
    for _ in range(2):
        net.zero_grad()
        net(x).backward(g)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        net.zero_grad()
        net(x).backward(g)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, 'device_time_total', None)
        if t is None:
            t = getattr(e, 'cuda_time_total', 0.0)
        st = getattr(e, 'self_device_time_total', None)
        if st is None:
            st = getattr(e, 'self_cuda_time_total', 0.0)
        if st <= 0:
            continue
        rows.append((st, e.count, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        f.write('# MiDaS fwd+bwd, %d images %dx%d, self device time per (op, shapes); total %.1f ms\n' % (
            a.images, a.H, a.W, tot / 1e3))
        for st, cnt, key, shp in rows[:120]:
            f.write('%9.1f us %5.1f%% x%-4d %-46s %s\n' % (st, 100 * st / tot, cnt, key[:46], shp))
    print(open(a.out).read()[:6000])


if __name__ == '__main__':
    main()
