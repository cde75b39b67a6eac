"""Oracle (test-only): the scene-flow field MLP and its Euler integration.

Functional restatement (weights passed as a dict keyed like the reference
state_dict) of
  * PeriodicEmbed.forward       /root/reference/networks/blocks.py:19-34
  * SceneFlowFieldNet.forward   /root/reference/networks/sceneflow_field.py:43-53
  * Conv2dBlock (1x1 conv + LeakyReLU(0.2))   networks/blocks.py:50-102
  * Model.forward_sf_net / forward_sf_net_multi_step
                                /root/reference/models/scene_flow_motion_field.py:346-367
  * kaiming init of the MLP     models/netinterface.py:66-74 with a=0.2
                                (scene_flow_motion_field.py:123)
"""

import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2


def frequencies(n_freq):
    # PeriodicEmbed(max_freq=N, N_freq=N): linspace(1, N + 1, N)  (blocks.py:24)
    return torch.linspace(1, n_freq + 1, steps=n_freq)


def periodic_embed(v, n_freq):
    """[B,C,H,W] -> [B, C*(1+2*n_freq), H, W]; order v, cos(f0 v).., sin(f0 v).."""
    if n_freq == 0:
        return v
    parts = [v]
    freqs = frequencies(n_freq)
    for fn in (torch.cos, torch.sin):
        for f in freqs:
            parts.append(fn(f * v))
    return torch.cat(parts, 1)


def layer_dims(n_freq_xyz=16, n_freq_t=16, width=256, n_hidden=4, out_dim=3, time_dependent=True):
    c_in = 3 + 6 * n_freq_xyz + ((1 + 2 * n_freq_t) if time_dependent else 0)
    return [c_in] + [width] * (n_hidden + 1) + [out_dim]


def init_params(seed=0, **kw):
    """kaiming_normal_(a=0.2, fan_in) weights, zero bias, reference key names."""
    dims = layer_dims(**kw)
    g = torch.Generator().manual_seed(seed)
    gain = math.sqrt(2.0 / (1 + LRELU_SLOPE ** 2))
    sd = {}
    for i in range(len(dims) - 1):
        std = gain / math.sqrt(dims[i])
        sd['convs.%d.conv.weight' % i] = torch.randn(dims[i + 1], dims[i], 1, 1, generator=g) * std
        sd['convs.%d.conv.bias' % i] = torch.zeros(dims[i + 1])
    return sd


def mlp_forward(sd, x, t=None, n_freq_xyz=16, n_freq_t=16):
    """SceneFlowFieldNet.forward: x [B,3,H,W], t [B,1,H,W] -> [B,3,H,W]."""
    x = x.contiguous()
    feat = periodic_embed(x, n_freq_xyz)
    if t is not None:
        feat = torch.cat([periodic_embed(t, n_freq_t), feat], 1)
    n_layers = len([k for k in sd if k.endswith('conv.weight')])
    h = feat
    for i in range(n_layers):
        h = F.conv2d(h, sd['convs.%d.conv.weight' % i], sd['convs.%d.conv.bias' % i])
        if i < n_layers - 1:
            h = F.leaky_relu(h, LRELU_SLOPE)
    return h


def sf_eval(sd, p, ts, sf_mag_div, **kw):
    """forward_sf_net: one evaluation, output divided by sf_mag_div."""
    return mlp_forward(sd, p, ts, **kw) / sf_mag_div


def sf_multi_step(sd, p, ts, time_step, steps, sf_mag_div, **kw):
    """forward_sf_net_multi_step: Euler advection, returns the summed flow."""
    acc = 0
    for _ in range(steps):
        sf = sf_eval(sd, p, ts, sf_mag_div, **kw)
        acc = acc + sf
        p = p + sf
        ts = ts + time_step
    return acc
