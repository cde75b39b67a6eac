import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # diagnostic, not a test: which hidden units flip LeakyReLU sign between devices
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
from oracle import sceneflow_mlp as M
from dvd_hip.networks.sceneflow_field import SceneFlowFieldNet
B,H,W = 2,24,40
sd = M.init_params(seed=3)
g = torch.Generator().manual_seed(B * 100 + W)
for k in sd:
    if k.endswith('bias'):
        sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
x = 3.0 * torch.randn(B, 3, H, W, generator=g)
tt = torch.rand(B, 1, 1, 1, generator=g).expand(B, 1, H, W).contiguous()
up = torch.randn(B, 3, H, W, generator=g)
sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
xr = x.clone().requires_grad_(True)
yr = M.mlp_forward(sdr, xr, tt)
(yr * up).sum().backward()
net = SceneFlowFieldNet(net_width=256, n_layers=4, time_dependent=True, N_freq_xyz=16, N_freq_t=16)
net.load_state_dict(sd); net = net.cuda()
for rep in range(3):
    xg = x.cuda().requires_grad_(True)
    yg = net(xg, tt.cuda())
    (yg * up.cuda()).sum().backward()
    e = (xg.grad.cpu() - xr.grad).abs()
    print('rep', rep, 'max err', float(e.max()), 'max ref', float(xr.grad.abs().max()))
    idx = torch.nonzero(e > 1e-3 * xr.grad.abs().max())
    print('n bad', len(idx))
    flat = (idx[:,0]*H*W + idx[:,2]*W + idx[:,3])
    print('bad pixels (linear idx, tile, chan, |x|):', [(int(f), int(f)//64, int(i[1]), float(x[i[0],i[1],i[2],i[3]])) for f,i in list(zip(flat, idx))[:20]])
    for k, p in net.named_parameters():
        ge = (p.grad.cpu() - sdr[k].grad).abs().max() / sdr[k].grad.abs().max()
        if rep == 0: print(k, float(ge))
    net.zero_grad()
