"""Autograd state of one MiDaS forward (with grad) per chunk size: what keeping a chunk's activations alive would cost."""
import json
import sys

import torch

sys.path.insert(0, 'dynamic-video-depth_amd')
from dvd_hip.third_party.MiDaS import MidasNet, calibrate_head_for_random_init  # noqa: E402

net = calibrate_head_for_random_init(MidasNet()).cuda().eval()
for p in net.parameters():
    p.requires_grad_(True)
for n in (4, 8, 16):
    x = torch.rand(n, 3, 384, 672, device='cuda')
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.enable_grad():
        d = net(x)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base
    peak = torch.cuda.max_memory_allocated() - base
    d.backward(torch.ones_like(d))
    torch.cuda.synchronize()
    peak_b = torch.cuda.max_memory_allocated() - base
    print(json.dumps({'images': n, 'held_after_forward_GB': held / 2 ** 30, 'peak_forward_GB': peak / 2 ** 30,
                      'peak_with_backward_GB': peak_b / 2 ** 30}))
    del d
    net.zero_grad(set_to_none=True)
