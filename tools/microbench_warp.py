#!/usr/bin/env python
"""Micro-benchmark of the fused warp+loss kernel at BASELINE config-2 size
(48 pairs, 384x672).  Prints achieved algorithmic HBM GB/s (52 B / pixel-pair,
SURVEY.md section 8d) from torch CUDA events on the launch stream."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
from dvd_hip import ops, synthetic  # noqa: E402

CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=48)
    ap.add_argument('--H', type=int, default=384)
    ap.add_argument('--W', type=int, default=672)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--fwd_only', action='store_true')
    ap.add_argument('--direct', action='store_true', help='global-atomics reference variant')
    ap.add_argument('--tiles', action='store_true', help='the tile kernel of rounds 2-5 instead of the strip kernel')
    ap.add_argument('--strip_rows', type=int, default=0, help='rows per unit of the strip kernel (0 = automatic)')
    ap.add_argument('--strip_shape', type=int, default=0, help='strip shape index (csrc/warp_strip.hip kStripShapes)')
    ap.add_argument('--tile', type=int, default=-1, help='force tile shape index')
    ap.add_argument('--px', type=int, default=0, help='pixels per thread-step (0 auto, 2, 4)')
    ap.add_argument('--flow_sigma', type=float, default=3.0)
    ap.add_argument('--smooth_flow', action='store_true', help='constant flow per pair instead of iid noise')
    ap.add_argument('--calm_border', type=int, default=0, help='zero the flow within this many pixels of the image border')
    a = ap.parse_args()
    B, H, W = a.B, a.H, a.W
    ops.warp_loss_select(variant='direct' if a.direct else ('tiles' if a.tiles else 'tiled'), tile=a.tile, px=a.px,
                         strip_rows=a.strip_rows, strip_shape=a.strip_shape)
    batch = synthetic.make_batch(B, H, W, device='cuda', with_images=False)
    batch['flow_1_2'] = batch['flow_1_2'] * (a.flow_sigma / 3.0)
    if a.smooth_flow:
        batch['flow_1_2'] = batch['flow_1_2'][:, :1, :1].expand(B, H, W, 2).contiguous()
    if a.calm_border:
        c = a.calm_border
        batch['flow_1_2'][:, :c] = 0
        batch['flow_1_2'][:, -c:] = 0
        batch['flow_1_2'][:, :, :c] = 0
        batch['flow_1_2'][:, :, -c:] = 0
    d1, d2 = synthetic.make_depths(B, H, W, device='cuda')
    sf = synthetic.make_scene_flow(B, H, W, device='cuda')
    cams = {k: batch[k] for k in CAM_KEYS}
    cfg = ops.warp_cfg(B, H, W)
    out = (torch.empty(4, device='cuda'), torch.empty_like(d1), torch.empty_like(d2), torch.empty_like(sf))
    grads = not a.fwd_only
    for _ in range(5):
        ops.warp_loss_fused(cfg, d1, d2, batch['flow_1_2'], batch['mask_2'], sf, cams, grads=grads, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.warp_loss_fused(cfg, d1, d2, batch['flow_1_2'], batch['mask_2'], sf, cams, grads=grads, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    npx = B * H * W
    bytes_alg = npx * (52 if grads else 32)
    print(json.dumps({'kernel': 'warp_loss_fused' if grads else 'warp_loss_fwd', 'B': B, 'H': H, 'W': W,
                      'ms_per_call_incl_memset_and_reduce': ms, 'algorithmic_bytes': bytes_alg,
                      'GBps': bytes_alg / ms / 1e6, 'frac_of_8TBps': bytes_alg / ms / 1e6 / 8000.0,
                      'smooth_flow': a.smooth_flow, 'direct': a.direct, 'tiles': a.tiles, 'strip_rows': a.strip_rows, 'strip_shape': a.strip_shape, 'tile': a.tile,
                      'flow_sigma': a.flow_sigma}))


if __name__ == '__main__':
    main()
