"""SceneFlowFieldNet on the fused gfx950 MLP kernels.

Drop-in for the reference class of the same name
(/root/reference/networks/sceneflow_field.py:20-53): same constructor
arguments, same `forward(x, t=None)` contract ([B,3,H,W], [B,1,H,W] ->
[B,3,H,W]) and the same state_dict keys `convs.{0..5}.conv.{weight,bias}`
(the 1x1 Conv2d parameters of networks/blocks.py:50-102), so a reference
checkpoint loads unchanged.  The arithmetic runs in
dvd_hip/csrc/sf_mlp.hip (periodic embedding + six 1x1 convs + LeakyReLU fused,
fp32 MFMA); autograd is provided by `_FusedMLP`, which calls the backward
kernels.  Only the configuration the reference Model builds is supported by
the kernels: width 256, four hidden layers, 3 outputs, LeakyReLU(0.2), no norm
(models/scene_flow_motion_field.py:107).
"""
import torch
from torch import nn

from .. import ops


class _Conv1x1(nn.Module):
    """Holds `conv` so that parameter names match the reference's Conv2dBlock."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, 1, 1, padding=0, bias=True)


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, t, *params):
        k = net.kernels(x.device)
        k.pack(params[0::2], params[1::2])
        B, _, H, W = x.shape
        n_pix = B * H * W
        need_grad = any(ctx.needs_input_grad)
        stash = k.new_stash(n_pix) if need_grad else None
        out = torch.empty_like(x.contiguous())
        k.forward(x, t, 0.0, 1.0, sf_out=out, stash=stash)
        ctx.net, ctx.stash, ctx.shape = net, stash, (B, H, W)
        ctx.param_shapes = [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, g_out):
        net, (B, H, W) = ctx.net, ctx.shape
        k = net.kernels(g_out.device)
        n_pix = B * H * W
        g_out = g_out.contiguous()
        dev = g_out.device
        gstash = k.new_gstash(n_pix)
        g_x = torch.empty(B, 3, H, W, device=dev, dtype=torch.float32)
        dims = [k.c_in] + [256] * 5
        gW = [torch.zeros(256 if i < 5 else 3, dims[i], device=dev) for i in range(6)]
        gb = [torch.zeros(256 if i < 5 else 3, device=dev) for i in range(6)]
        k.backward_dx(ctx.stash, 1.0, g_out, g_x, gstash, gW[5], gb[5], (B, H, W))
        k.backward_dw(ctx.stash, gstash, n_pix, gW[:5], gb[:5])
        grads = []
        for i in range(6):
            grads.append(gW[i].view(ctx.param_shapes[2 * i]))
            grads.append(gb[i])
        ctx.stash = None
        return (None, g_x, None) + tuple(grads)


class SceneFlowFieldNet(nn.Module):
    def __init__(self, time_dependent=True, N_freq_xyz=0, N_freq_t=0, output_dim=3, net_width=32, n_layers=3,
                 activation='lrelu', norm='none'):
        super().__init__()
        if output_dim != 3 or net_width != 256 or n_layers != 4 or activation != 'lrelu' or norm != 'none':
            raise NotImplementedError('the fused HIP scene-flow MLP implements the configuration the reference '
                                      'Model uses: net_width=256, n_layers=4, output_dim=3, lrelu, norm none')
        c_xyz = 3 + 3 * 2 * N_freq_xyz
        c_t = 1 + 2 * N_freq_t
        c_in = c_xyz + c_t if time_dependent else c_xyz
        layers = [_Conv1x1(c_in, net_width)]
        layers += [_Conv1x1(net_width, net_width) for _ in range(n_layers)]
        layers.append(_Conv1x1(net_width, output_dim))
        self.convs = nn.Sequential(*layers)
        self.time_dependent = time_dependent
        self.n_freq_xyz, self.n_freq_t = N_freq_xyz, N_freq_t
        self._kernels = {}

    def kernels(self, device, stash_f16=False):
        key = (str(device), bool(stash_f16))
        if key not in self._kernels:
            self._kernels[key] = ops.SceneFlowMLPKernels(device, self.n_freq_xyz, self.n_freq_t, self.time_dependent,
                                                         stash_f16=stash_f16)
        return self._kernels[key]

    def parameter_list(self):
        out = []
        for blk in self.convs:
            out += [blk.conv.weight, blk.conv.bias]
        return out

    def forward(self, x, t=None):
        if t is None and self.time_dependent:
            raise ValueError
        return _FusedMLP.apply(self, x, t if self.time_dependent else None, *self.parameter_list())
