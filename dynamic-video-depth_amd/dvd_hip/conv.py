"""Convolutions of the depth networks that run on hand-written HIP kernels.

`GroupedConv3x3C8` is `nn.Conv2d(C, C, 3, padding=1, groups=C // 8, bias=False)` -- the
`conv2` of the ResNeXt-101 32x8d stage-1 bottlenecks inside the MiDaS encoder (reference:
third_party/midas_blocks.py:35-50; torchvision resnet.py Bottleneck with groups=32,
width_per_group=8).  Same parameter name and shape (`weight [C, 8, 3, 3]`), so
state_dicts interchange.  Forward, backward-data and backward-weight go through
`dvd_gconv3x3_c8_*` (csrc/gconv.hip); MIOpen's immediate mode serves the backward of this
shape at 0.3 TFLOP/s (profiles/r01_depthnet_profile.txt), which made three small
convolutions 38 % of the depth net's time.

Tensors that the kernels do not cover (CPU tensors of the oracle/tests, other dtypes)
take `F.conv2d`; on a GPU in fp32 the HIP path is the one that runs.
"""
import ctypes
import os as _os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .ops import _p, _stream, _workspace, amax_of, chansum, known_amax, new_scalar, set_amax

# Same-box A/B measurement switches (previous-generation kernels / MIOpen against the kernels in use), read ONCE at
# import from DVD_AB="gconv32,no_bnfuse,...".  Not product configuration: every default is the fastest measured path.
#   gconv32     32-per-group 3x3 on round 1's fp32-MFMA kernels (csrc/gconv32.hip) instead of the grouped xconv path
#   no_c16      16-per-group 3x3 on MIOpen instead of paired groups on the 32-per-group kernels
#   no_xwgrad3  weight gradients on the exact-fp32 MFMA kernel (csrc/xwgrad.hip);  no_xwgrad: on MIOpen
#   no_bnfuse   BatchNorm (+ residual, ReLU) as a separate pass after the convolution
#   no_xconv    dense convolutions on MIOpen
#   no_alias    gradient joins of residual blocks by autograd's accumulation (ATen add) instead of the backward-data epilogue
#   no_maskfuse the ReLU mask of a BatchNorm+ReLU site always in the site's own mask pass (never in its consumer's epilogue)
#   no_packplan every captured graph packs its weights itself (two launches per weight tensor and graph)
#   no_chansum  bias gradients by an ATen sum (and max|gy| by dvd_amax) instead of one dvd_chansum read
#   no_s2       stride-2 3x3 convolutions as rounds 2-5 ran them: the stride-1 kernel's output sub-sampled, the gradient
#               zero-interleaved in HBM (csrc/pool.hip) in front of the stride-1 backward-data kernel
#   rowsum      (opt-in experiment) a pre-masked site takes its per-channel sums from dvd_xwgrad1s_rowsum and max|g| from the
#               consumer's epilogue instead of running its sum pass
AB = {k: False for k in ('gconv32', 'no_c16', 'no_xwgrad3', 'no_xwgrad', 'no_bnfuse', 'no_xconv', 'no_alias', 'no_maskfuse',
                         'no_chansum', 'no_packplan', 'no_s2',
                         'rowsum')}
for _k in filter(None, _os.environ.get('DVD_AB', '').split(',')):
    if _k not in AB:
        raise RuntimeError('DVD_AB: unknown switch %r (known: %s)' % (_k, ', '.join(sorted(AB))))
    AB[_k] = True


# ---- fp16 activation storage (BASELINE configs[4]) ------------------------------------------------------------------
# Activations between the encoder stem and the depth head may be torch.float16 tensors: the same autograd Functions then call the
# "_h" / "_t" entry points (csrc/xconv.hip IN16 / OUT16, csrc/xwgrad3.hip H16, the templated helper kernels).  fp16 gradients
# carry the step's loss scale S (csrc/a16.hip); every PARAMETER gradient is multiplied by 1 / S where it is produced, so the
# flat gradient buffers hold true fp32 gradients.  The loss-scale state (16 floats on the device) belongs to the model that is
# stepping; it is registered here because the backward Functions need its address.
ACT_DTYPES = (torch.float32, torch.float16)
GRAD_SCALE = {'state': None}


def set_grad_scale_state(state):
    """state: the 16-float device tensor of dvd_gscale_* (or None).  Its address is baked into captured HIP graphs, so a model
    allocates it once and keeps it for its lifetime."""
    GRAD_SCALE['state'] = state


def _is16(t):
    return t.dtype == torch.float16


def _gs(i):
    """1-element view of the loss-scale state: 1 = 1 / S (out_scale of parameter gradients), 3 = observed max |S g|."""
    st = GRAD_SCALE['state']
    if st is None:
        raise RuntimeError('fp16 gradients need a loss-scale state (conv.set_grad_scale_state; Model does it with --act_fp16)')
    return st[i:i + 1]


def _fwd_monitor():
    """Slot [6] of the loss-scale state: max |activation| that the fp16-output convolution epilogues and the depth head fold
    in during the forward pass (NaN as +Inf).  dvd_gscale_end skips the step when an activation left fp16's range: its Inf
    would otherwise reach the weight gradients (Inf x 0 = NaN) without any GRADIENT monitor seeing it.  None
    without a state (inference with fp16 activations outside a training model)."""
    st = GRAD_SCALE['state']
    return None if st is None else st[6:7]


class _BnRelu(torch.autograd.Function):
    """y = relu(bn_eval(x) (+ residual)); see csrc/bnrelu.hip."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, mean, var, eps, relu):
        x = x.contiguous()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        if residual is not None:
            residual = residual.contiguous()
        y = torch.empty_like(x)
        lib = _lib.load()
        # fp32 storage: max|y| from the same pass (the 1x1 convolution that follows takes its operand scale from it; until
        # round 6 a dvd_amax pass over y: 18 of them per step behind the 8- and 16-per-group and the strided convolutions)
        y_amax = None if _is16(x) else new_scalar(x.device)
        _lib.check(lib.dvd_bnrelu_fwd_m(_p(x), _p(residual), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), _p(y),
                                        int(_is16(x)), N, C, HW, int(relu), _p(y_amax), _stream()), 'dvd_bnrelu_fwd')
        ctx.save_for_backward(x, y if relu else None, gamma, mean, var)
        ctx.cfg = (N, C, HW, float(eps), int(relu), residual is not None)
        if y_amax is not None:
            ctx.mark_non_differentiable(y_amax)
        return y, y_amax

    @staticmethod
    def backward(ctx, gy, _g_amax=None):
        x, y, gamma, mean, var = ctx.saved_tensors
        N, C, HW, eps, relu, has_res = ctx.cfg
        gy = gy.contiguous()
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        gr = torch.empty_like(x) if (has_res and need[1]) else None
        gg = torch.empty_like(gamma) if need[2] else None
        gb = torch.empty_like(gamma) if need[3] else None
        lib = _lib.load()
        ws = _workspace(lib.dvd_bnrelu_bwd_workspace_bytes(N, C, HW), x.device)
        h16 = _is16(gy)
        # fp32: max|gx| (and max|g| for the residual branch) come with the pass; the convolutions' backward found neither
        # attached and ran a dvd_amax pass each (12 per step)
        gx_amax = new_scalar(gy.device) if (gx is not None and not h16) else None
        gr_amax = new_scalar(gy.device) if (gr is not None and not h16) else None
        _lib.check(lib.dvd_bnrelu_bwd_m(_p(gy), _p(y), _p(x), _p(gamma), _p(mean), _p(var), eps, _p(gx), _p(gr), _p(gg),
                                        _p(gb), _p(ws), ctypes.c_size_t(ws.numel()), int(h16), _p(_gs(1)) if h16 else None, N, C,
                                        HW, relu, _p(_gs(3)) if h16 else _p(gr_amax), _p(gx_amax), _stream()), 'dvd_bnrelu_bwd')
        if gx_amax is not None:
            set_amax(gx, gx_amax)
        if gr_amax is not None:
            set_amax(gr, gr_amax)
        return gx, gr, gg, gb, None, None, None, None


def bn_eval_relu(bn, x, residual=None, relu=True):
    """relu(bn(x) (+ residual)) for an nn.BatchNorm2d in eval mode (running statistics; gamma / beta keep
    their gradients) on the fused HIP kernel; anything else (training-mode BN, CPU, other dtypes) takes the
    ATen ops the reference uses."""
    if (not bn.training and bn.track_running_stats and x.is_cuda and x.dtype in ACT_DTYPES):
        if bn.affine:
            gamma, beta = bn.weight, bn.bias
        else:                                   # hourglass inception blocks: BatchNorm2d(affine=False)
            gamma, beta = torch.ones_like(bn.running_mean), torch.zeros_like(bn.running_mean)
        y, y_amax = _BnRelu.apply(x, residual, gamma, beta, bn.running_mean, bn.running_var, bn.eps, relu)
        return set_amax(y, y_amax)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _ToHalf(torch.autograd.Function):
    """The fp32 -> fp16 boundary behind the encoder stem: forward casts; backward hands the fp32 stem the TRUE gradient, i.e.
    the fp16 gradient times 1 / (loss scale) (dvd_cast_scale_f32), so the stem's parameter gradients need no special case."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if g.numel() % 4:
            return g.float() * _gs(1)
        out = torch.empty(g.shape, device=g.device, dtype=torch.float32)
        _lib.check(_lib.load().dvd_cast_scale_f32(_p(g), 1, _p(out), ctypes.c_longlong(g.numel()), _p(_gs(1)), _stream()),
                   'dvd_cast_scale_f32')
        return out


def to_half(x):
    return _ToHalf.apply(x)


class _Head1x1(torch.autograd.Function):
    """`Conv2d(C, 1, 1)(relu(x))` of the depth head (third_party/MiDaS.py:192-194) with fp16 (or fp32) features and fp32
    output: the fp16 / fp32 boundary at the network's end.  Backward starts the pass's loss scale (dvd_gscale_begin from
    max|g_out| and max|w|), then writes the feature gradient times S in x's storage and the fp32 weight / bias gradients
    (csrc/a16.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu_in):
        x = x.contiguous()
        N, C, H, W = x.shape
        y = torch.empty(N, 1, H, W, device=x.device, dtype=torch.float32)
        # (fp16 features: the head output's max|y| goes to the forward monitor of the overflow guard, state[6])
        _lib.check(_lib.load().dvd_head1x1_fwd(_p(x), int(_is16(x)), _p(weight), _p(bias), _p(y),
                                               _p(_fwd_monitor()) if _is16(x) else None, N, C, H * W, int(bool(relu_in)),
                                               _stream()), 'dvd_head1x1_fwd')
        ctx.save_for_backward(x, weight)
        ctx.cfg = (bool(relu_in), bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .ops import amax
        x, weight = ctx.saved_tensors
        relu_in, has_bias = ctx.cfg
        gy = gy.contiguous().float()
        N, C, H, W = x.shape
        lib = _lib.load()
        h16 = _is16(x)
        state = None
        if h16:
            state = GRAD_SCALE['state']
            if state is None:
                raise RuntimeError('fp16 gradients need a loss-scale state (conv.set_grad_scale_state)')
            _lib.check(lib.dvd_gscale_begin(_p(state), _p(amax(gy)), _p(weight), C, _stream()), 'dvd_gscale_begin')
        gx = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty(1, device=x.device, dtype=torch.float32) if has_bias else None
        ws = _workspace(lib.dvd_head1x1_bwd_workspace_bytes(C), x.device)
        _lib.check(lib.dvd_head1x1_bwd(_p(x), int(h16), _p(weight), _p(gy), _p(state), _p(gx), _p(gw), _p(gb), _p(ws),
                                       ctypes.c_size_t(ws.numel()), N, C, H * W, int(relu_in), _stream()), 'dvd_head1x1_bwd')
        return gx, gw, gb, None


class _DepthTail(torch.autograd.Function):
    """`10000 / torch.clamp(relu(v), min=1e-2)` (third_party/MiDaS.py:192-195,240-242) as one HIP pass each way
    (csrc/elementwise.hip dvd_depth_tail_*): the forward value is torch's (reciprocal, then x 10000), the backward autograd's formula."""

    @staticmethod
    def forward(ctx, v):
        v = v.contiguous()
        out = torch.empty_like(v)
        _lib.check(_lib.load().dvd_depth_tail_fwd(_p(v), _p(out), v.numel(), _stream()), 'dvd_depth_tail_fwd')
        ctx.save_for_backward(v)
        return out

    @staticmethod
    def backward(ctx, g):
        v, = ctx.saved_tensors
        g = g.contiguous()
        gv = torch.empty_like(v)
        _lib.check(_lib.load().dvd_depth_tail_bwd(_p(v), _p(g), _p(gv), v.numel(), _stream()), 'dvd_depth_tail_bwd')
        return gv


def depth_tail(v):
    """10000 / clamp(relu(v), 1e-2): the HIP pass for fp32 GPU tensors, the reference's ATen expression otherwise."""
    if v.is_cuda and v.dtype == torch.float32:
        return _DepthTail.apply(v)
    return 10000 / torch.clamp(F.relu(v), min=1e-2)


def head1x1(conv, x, relu_in=True):
    """conv(relu(x)) for an nn.Conv2d(C, 1, 1) on the boundary kernel (GPU tensors, C <= 64, H * W % 4 == 0)."""
    return _Head1x1.apply(x, conv.weight, conv.bias, relu_in)


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw, align_corners):
        x = x.contiguous()
        N, C, H, W = x.shape
        Ho, Wo = out_hw
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype)
        lib = _lib.load()
        _lib.check(lib.dvd_upsample_bilinear_fwd_t(_p(x), _p(y), int(_is16(x)), N * C, H, W, Ho, Wo, int(align_corners),
                                                   _stream()), 'dvd_upsample_bilinear_fwd')
        ctx.shape, ctx.align = (N, C, H, W, Ho, Wo), int(align_corners)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W, Ho, Wo = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(N, C, H, W, device=gy.device, dtype=gy.dtype)
        lib = _lib.load()
        _lib.check(lib.dvd_upsample_bilinear_bwd_t(_p(gy), _p(gx), int(_is16(gy)), N * C, H, W, Ho, Wo, ctx.align, _stream()),
                   'dvd_upsample_bilinear_bwd')
        return gx, None, None


def upsample_bilinear2x(x, align_corners):
    """F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=...) of the MiDaS decoder."""
    if x.is_cuda and x.dtype in ACT_DTYPES:
        y = _UpsampleBilinear.apply(x, (2 * x.shape[2], 2 * x.shape[3]), bool(align_corners))
        return set_amax(y, known_amax(x))            # a convex combination never exceeds the largest input magnitude
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=align_corners)


class _GConv3x3C8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()
        w = w.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        lib = _lib.load()
        _lib.check(lib.dvd_gconv3x3_c8_fwd_t(_p(x), _p(w), _p(y), int(_is16(x)), N, C, H, W, _stream()), 'dvd_gconv3x3_c8_fwd')
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        N, C, H, W = x.shape
        lib = _lib.load()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(lib.dvd_gconv3x3_c8_bwd_data_t(_p(gy), _p(w), _p(gx), int(_is16(gy)), N, C, H, W, _stream()),
                       'dvd_gconv3x3_c8_bwd_data')
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            nws = lib.dvd_gconv3x3_c8_wgrad_workspace_bytes(N, C, H, W)
            ws = _workspace(nws, x.device)
            h16 = _is16(gy)
            _lib.check(lib.dvd_gconv3x3_c8_bwd_weight_t(_p(x), _p(gy), _p(gw), 0, _p(ws), ctypes.c_size_t(ws.numel()), int(h16),
                                                        _p(_gs(1)) if h16 else None, N, C, H, W, _stream()),
                       'dvd_gconv3x3_c8_bwd_weight')
        return gx, gw


class _GConv3x3C32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()
        w = w.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        lib = _lib.load()
        ws = _workspace(lib.dvd_gconv3x3_c32_workspace_bytes(N, C, H, W), x.device)
        _lib.check(lib.dvd_gconv3x3_c32_fwd(_p(x), _p(w), _p(y), _p(ws), ctypes.c_size_t(ws.numel()), N, C, H, W,
                                            _stream()), 'dvd_gconv3x3_c32_fwd')
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        N, C, H, W = x.shape
        lib = _lib.load()
        ws = _workspace(lib.dvd_gconv3x3_c32_workspace_bytes(N, C, H, W), x.device)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(lib.dvd_gconv3x3_c32_bwd_data(_p(gy), _p(w), _p(gx), _p(ws), ctypes.c_size_t(ws.numel()), N, C, H,
                                                     W, _stream()), 'dvd_gconv3x3_c32_bwd_data')
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            _lib.check(lib.dvd_gconv3x3_c32_bwd_weight(_p(x), _p(gy), _p(gw), 0, _p(ws), ctypes.c_size_t(ws.numel()),
                                                       N, C, H, W, _stream()), 'dvd_gconv3x3_c32_bwd_weight')
        return gx, gw


def gconv3x3_c32(x, weight):
    """y = conv2d(x, weight, padding=1, groups=C // 32) for weight [C, 32, 3, 3]."""
    if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32:
        return _GConv3x3C32.apply(x, weight)
    return F.conv2d(x, weight, None, 1, 1, 1, x.shape[1] // 32)


class GroupedConv3x3C32(nn.Conv2d):
    """Drop-in for nn.Conv2d(C, C, 3, stride=1, padding=1, groups=C // 32, bias=False) (fp32 MFMA kernels)."""

    def __init__(self, channels):
        if channels % 32:
            raise ValueError('GroupedConv3x3C32 needs a multiple of 32 channels')
        super().__init__(channels, channels, 3, stride=1, padding=1, groups=channels // 32, bias=False)

    def forward(self, x):
        if x.is_cuda and x.dtype in ACT_DTYPES and not AB['gconv32']:
            # the grouped split-operand MFMA kernels (csrc/xconv.hip, csrc/xwgrad3.hip): 0.097 ms forward / 0.37 ms backward per
            # 16-image call at [1024, 24, 42] against 0.156 / 0.42 ms of the fp32-MFMA kernels (tools/microbench_gx.py)
            return _xconv(x, self.weight, None, None, False, False, self.groups)
        return gconv3x3_c32(x, self.weight)


# ---- the ResNeXt stem natively (round 4) ------------------------------------------------------------------------------
# `conv1 = Conv2d(3, 64, 7, stride 2, padding 3)` + `bn1` + ReLU + `MaxPool2d(3, 2, 1)` (torchvision's ResNet stem, reached
# through third_party/midas_blocks.py:35-45).  Round 3 left all four to ATen / MIOpen.  A stride-2 7x7 convolution is a
# STRIDE-1 convolution over the 2x2 space-to-depth image: with x'[4c + 2py + px][Y][X] = x[c][2Y + py][2X + px] and
# ky - 3 = 2a + py (a in -2..1), out[y][x] = sum w[c][ky][kx] x'[4c + 2py + px][y + a][x + b] -- a 4x4 kernel, embedded in a
# 5x5 "same" one (zero taps at a = 2 / b = 2).  That runs on the kernels this package already has: xconv (csrc/xconv.hip) with
# the eval-mode BatchNorm + ReLU in its epilogue, and the exact-fp32 weight gradient (csrc/xwgrad.hip, KS = 5); the image needs
# no gradient.  The rearranged weight is a gather of the module's [64,3,7,7] parameter, so its gradient flows back through
# autograd's index backward and the state_dict keeps the reference's key and shape.
_S2D_IDX = {}


def _s2d_index(device):
    key = str(device)
    if key not in _S2D_IDX:
        idx = torch.full((2, 2, 5, 5), 49, dtype=torch.long)
        for py in range(2):
            for a in range(-2, 3):
                ky = 2 * a + py + 3
                if not 0 <= ky <= 6:
                    continue
                for px in range(2):
                    for b in range(-2, 3):
                        kx = 2 * b + px + 3
                        if 0 <= kx <= 6:
                            idx[py, px, a + 2, b + 2] = ky * 7 + kx
        _S2D_IDX[key] = idx.reshape(-1).to(device)
    return _S2D_IDX[key]


def s2d_weight(w):
    """[Cout, C, 7, 7] (stride 2, padding 3) -> the equivalent [Cout, 4C, 5, 5] (stride 1, padding 2) over pixel_unshuffle(x, 2)."""
    Co, C = w.shape[0], w.shape[1]
    w_ext = torch.cat([w.reshape(Co, C, 49), w.new_zeros(Co, C, 1)], 2)
    return w_ext.index_select(2, _s2d_index(w.device)).reshape(Co, C * 4, 5, 5)


def is_resnet_stem(conv, bn):
    """The 7x7 / stride 2 / padding 3 convolution + eval-mode BatchNorm that stem_conv_bn_relu implements (any other stem
    stays on ATen: MidasNet.forward checks this before it takes the native path -- ADVICE round 4)."""
    return (isinstance(conv, torch.nn.Conv2d) and isinstance(bn, torch.nn.BatchNorm2d) and conv.kernel_size == (7, 7) and
            conv.stride == (2, 2) and conv.padding == (3, 3) and conv.groups == 1 and conv.dilation == (1, 1) and
            not bn.training and bn.track_running_stats)


def stem_conv_bn_relu(conv, bn, x):
    """relu(bn(conv(x))) for the 7x7 / stride 2 / padding 3 stem convolution and its eval-mode BatchNorm, GPU fp32 tensors."""
    if not is_resnet_stem(conv, bn):
        raise RuntimeError('stem_conv_bn_relu: not the ResNet stem (%r)' % (conv,))
    H, W = x.shape[2], x.shape[3]
    if H % 2 or W % 2:                    # the zero row / column the convolution's padding would have supplied
        x = F.pad(x, (0, W % 2, 0, H % 2))
    x4 = F.pixel_unshuffle(x, 2).contiguous()
    w5 = s2d_weight(conv.weight)
    gamma, beta = (bn.weight, bn.bias) if bn.affine else (None, None)
    out_site = _Site()
    y, y_amax = _XConvBn.apply(x4, amax_of(x4), w5, conv.bias, gamma, beta, bn.running_mean, bn.running_var, bn.eps, None, True,
                               1, False, None, out_site)
    y._dvd_site = out_site
    return set_amax(y, y_amax)


class _MaxPool3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on csrc/pool.hip (ATen's tie rule; deterministic gather backward).  to_half: the output is
    written as fp16 -- the fp32 / fp16 boundary of fp16 activation storage, fused into the pooling; the backward then hands
    the fp32 stem the TRUE gradient (fp16 gradient times 1 / loss scale)."""

    @staticmethod
    def forward(ctx, x, to_half):
        x = x.contiguous()
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=torch.float16 if to_half else torch.float32)
        idx = torch.empty(N, C, Ho, Wo, device=x.device, dtype=torch.uint8)
        _lib.check(_lib.load().dvd_maxpool3s2_fwd(_p(x), _p(y), int(bool(to_half)), _p(idx), ctypes.c_longlong(N * C), H, W,
                                                  _stream()), 'dvd_maxpool3s2_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        h16 = _is16(gy)
        gx = torch.empty(N, C, H, W, device=gy.device, dtype=torch.float32)
        _lib.check(_lib.load().dvd_maxpool3s2_bwd(_p(gy), int(h16), _p(idx), _p(gx), _p(_gs(1)) if h16 else None,
                                                  ctypes.c_longlong(N * C), H, W, _stream()), 'dvd_maxpool3s2_bwd')
        return gx, None


def maxpool3s2(x, to_half=False):
    y = _MaxPool3s2.apply(x, bool(to_half))
    return y if to_half else set_amax(y, known_amax(x))      # a maximum of inputs never exceeds the largest input magnitude


def add_bounded(a, b):
    """a + b; when both operands carry a max|.| scalar the sum's bound is their sum (one tiny kernel instead of a reduction
    pass over the sum when it feeds a convolution)."""
    y = a + b
    ka, kb = known_amax(a), known_amax(b)
    return set_amax(y, ka + kb) if (ka is not None and kb is not None) else y


class _Subsample2(torch.autograd.Function):
    """x[:, :, ::2, ::2] as a contiguous tensor on csrc/pool.hip (dvd_subsample2_*): one gather forward, one zero-interleaving
    pass backward (ATen: a strided copy; a fill and a strided copy)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, C, H, W = x.shape
        y = torch.empty(N, C, (H + 1) // 2, (W + 1) // 2, device=x.device, dtype=x.dtype)
        _lib.check(_lib.load().dvd_subsample2_fwd(_p(x), _p(y), int(_is16(x)), N * C, H, W, _stream()), 'dvd_subsample2_fwd')
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(N, C, H, W, device=gy.device, dtype=gy.dtype)
        _lib.check(_lib.load().dvd_subsample2_bwd(_p(gy), _p(gx), int(_is16(gy)), N * C, H, W, _stream()), 'dvd_subsample2_bwd')
        k = known_amax(gy)
        return gx if k is None else set_amax(gx, k)


def _subsample(t, st):
    """t[:, :, ::st, ::st] as a contiguous tensor; max|t| stays an upper bound of the result's."""
    if st == 2 and t.is_cuda and t.dtype in ACT_DTYPES and t.dim() == 4:
        y = _Subsample2.apply(t)
    else:
        y = t[:, :, ::st, ::st].contiguous()
    k = known_amax(t)                     # the producer's bound if there is one; else reduce the (st^2 x smaller) result
    return set_amax(y, k) if k is not None else y


class _AvgPool(torch.autograd.Function):
    """F.avg_pool2d(x, k, stride, pad) with PyTorch's defaults on csrc/pool.hip (dvd_avgpool_*; ATen's arithmetic)."""

    @staticmethod
    def forward(ctx, x, k, st, pad):
        x = x.contiguous()
        N, C, H, W = x.shape
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype)
        _lib.check(_lib.load().dvd_avgpool_fwd(_p(x), _p(y), int(_is16(x)), N * C, H, W, k, st, pad, _stream()), 'dvd_avgpool_fwd')
        ctx.cfg = (N, C, H, W, k, st, pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W, k, st, pad = ctx.cfg
        gy = gy.contiguous()
        gx = torch.empty(N, C, H, W, device=gy.device, dtype=gy.dtype)
        _lib.check(_lib.load().dvd_avgpool_bwd(_p(gy), _p(gx), int(_is16(gy)), N * C, H, W, k, st, pad, _stream()), 'dvd_avgpool_bwd')
        kg = known_amax(gy)                   # at most ceil(k / st)^2 windows of weight 1 / k^2 hold a pixel: |gx| <= max|gy|
        return (gx if kg is None else set_amax(gx, kg)), None, None, None


class AvgPool2d(nn.AvgPool2d):
    """Drop-in nn.AvgPool2d: GPU fp32 / fp16 tensors of the default configuration (floor mode, count_include_pad, square
    window, 2 * padding <= kernel) run on csrc/pool.hip -- the hourglass's AvgPool2d(2) (third_party/hourglass.py:60-158) and
    FCNUnet's AvgPool2d(3, 2, 1) (networks/FCNUnet.py:64); everything else, and CPU tensors, take ATen."""

    def forward(self, x):
        def one(v):
            return v if isinstance(v, int) else (v[0] if v[0] == v[1] else None)
        k, st, pad = one(self.kernel_size), one(self.stride), one(self.padding)
        if (x.is_cuda and x.dtype in ACT_DTYPES and x.dim() == 4 and None not in (k, st, pad) and not self.ceil_mode and
                self.count_include_pad and self.divisor_override is None and 1 <= k <= 7 and 2 * pad <= k):
            y = _AvgPool.apply(x, k, st, pad)
            kx = known_amax(x)                # an average of inputs (and padding zeros) never exceeds the largest magnitude
            return y if kx is None else set_amax(y, kx)
        return super().forward(x)


def _pair_groups_of_16(weight):
    """[C,16,3,3] (16 channels per group) -> the equivalent [C,32,3,3] with groups paired into blocks of 32
    channels and zeros off the 16x16 diagonal blocks (exact: the extra products are with 0.0)."""
    C = weight.shape[0]
    w = weight.view(C // 32, 2, 16, 16, 3, 3)
    w32 = weight.new_zeros(C // 32, 2, 16, 2, 16, 3, 3)
    w32[:, 0, :, 0] = w[:, 0]
    w32[:, 1, :, 1] = w[:, 1]
    return w32.view(C, 32, 3, 3)


class GroupedConv3x3C16(nn.Conv2d):
    """nn.Conv2d(C, C, 3, stride=1, padding=1, groups=C // 16, bias=False) (ResNeXt stage 2, stride-1 blocks)
    on the 32-channel MFMA kernels: two groups share one 32x32 tile with a block-diagonal weight.  MIOpen's
    immediate mode runs this shape per image as im2col + small GEMMs (~80 launches per call)."""

    def __init__(self, channels, stride=1):
        if channels % 32:
            raise ValueError('GroupedConv3x3C16 needs a multiple of 32 channels')
        super().__init__(channels, channels, 3, stride=stride, padding=1, groups=channels // 16, bias=False)

    def forward(self, x):
        st = self.stride[0]
        if x.is_cuda and x.dtype in ACT_DTYPES and not AB['no_c16'] and (st == 1 or not AB['gconv32']):
            if not AB['gconv32']:
                w32 = _pair_groups_of_16(self.weight)
                if st == 2 and xconv_s2_supported(x, w32, self.groups // 2):
                    return _xconv_s2(x, w32, None, self.groups // 2)
                y = _xconv(x, w32, None, None, False, False, self.groups // 2)
                # the stride-2 entry of stage 2: out[i][j] of a strided 'same' 3x3 is the stride-1 result at [s*i][s*j]
                # (4x the needed work, still half of MIOpen's per-image im2col + GEMM + col2im path)
                return y if st == 1 else _subsample(y, st)
            return gconv3x3_c32(x, _pair_groups_of_16(self.weight))
        return F.conv2d(x, self.weight, None, self.stride, 1, 1, self.groups)


def gconv3x3_c8(x, weight):
    """y = conv2d(x, weight, padding=1, groups=C // 8) for weight [C, 8, 3, 3]."""
    if x.is_cuda and x.dtype in ACT_DTYPES and weight.dtype == torch.float32:
        return _GConv3x3C8.apply(x, weight)
    return F.conv2d(x, weight, None, 1, 1, 1, x.shape[1] // 8)


class GroupedConv3x3C8(nn.Conv2d):
    """Drop-in for nn.Conv2d(C, C, 3, stride=1, padding=1, groups=C // 8, bias=False)."""

    def __init__(self, channels):
        if channels % 8:
            raise ValueError('GroupedConv3x3C8 needs a multiple of 8 channels')
        super().__init__(channels, channels, 3, stride=1, padding=1, groups=channels // 8, bias=False)

    def forward(self, x):
        return gconv3x3_c8(x, self.weight)


# ---------------------------------------------------------------------------------------
# Dense stride-1 convolutions on the split-operand MFMA kernels (csrc/xconv.hip; two fp16 terms per operand, csrc/dvd_split.h)

class _PackPlan(object):
    """Every packing a captured depth-net graph needs, made by TWO launches per optimiser step (dvd_xconv_pack_many) instead
    of two per weight tensor and per captured chunk (round 5: 964 launches, 6.3 ms of a 668 ms step).

    Eager calls of xconv_packed / xconv_packed_scaled (the warm-up passes before a capture) leave a REQUEST on the weight;
    `extend()` -- called by the model right before it opens a capture -- gives every requested packing a persistent buffer and
    a row of the device table; under capture xconv_packed then returns that buffer and launches nothing.  The buffers hold
    the packing of the weights' CURRENT values only after `ensure_current()`: the owner of the graphs calls it before every
    replay (it compares version counters / the fused Adam's epoch on the host and launches only after a change).
    Entries hold weak references: a discarded model's rows are dropped at the next extend()."""

    def __init__(self):
        self.requests = {}                    # id(weight) -> weak reference: weights with a '_dvd_pack_req' not taken yet
        self.entries = []
        self.items = None
        self.table = None
        self.key = None
        self.dirty = False
        self.launches = 0                     # (tests / bench read it)

    def request(self, weight, kind, groups, gamma=None, var=None, eps=0.0):
        planned = getattr(weight, '_dvd_plan', None)
        if planned is not None and kind in planned:
            e = planned[kind]
            if kind != 'Ts' or (e['gamma']() is gamma and e['var']() is var and e['eps'] == float(eps)):
                return
        req = getattr(weight, '_dvd_pack_req', None)
        if req is None:
            req = {}
            weight._dvd_pack_req = req
        req[kind] = (int(groups), gamma, var, float(eps))
        if id(weight) not in self.requests:
            import weakref
            key = id(weight)
            self.requests[key] = weakref.ref(weight, lambda _r, key=key, reqs=self.requests: reqs.pop(key, None))

    @staticmethod
    def _ref(t):
        import weakref
        return weakref.ref(t) if t is not None else (lambda: None)

    def extend(self):
        """Take the pending requests (not during a capture) and bring every buffer up to date."""
        if torch.cuda.is_current_stream_capturing():
            return
        lib = _lib.load()
        import weakref
        live = [e for e in self.entries if e['w']() is not None]
        if len(live) != len(self.entries):
            self.entries, self.dirty = live, True
        for ref in list(self.requests.values()):
            weight = ref()
            if weight is None:
                continue
            req = getattr(weight, '_dvd_pack_req', None) or {}
            planned = getattr(weight, '_dvd_plan', None)
            if planned is None:
                planned = {}
                weight._dvd_plan = planned
            Cout, Cin, KS, _ = weight.shape
            for kind, (groups, gamma, var, eps) in req.items():
                transposed = kind != 'F'
                nbytes = lib.dvd_xconv_packed_bytes(Cout, Cin * groups, KS, groups, int(transposed))
                if nbytes == 0 or not weight.is_contiguous() or weight.dtype != torch.float32:
                    continue
                old = planned.get(kind)
                if old is not None:
                    self.entries = [e for e in self.entries if e is not old]
                e = dict(w=weakref.ref(weight), kind=kind, groups=groups, gamma=self._ref(gamma), var=self._ref(var), eps=eps,
                         packed=torch.empty(nbytes, device=weight.device, dtype=torch.uint8), shape=(Cout, Cin * groups, KS))
                planned[kind] = e
                self.entries.append(e)
                self.dirty = True
            weight._dvd_pack_req = None
        self.requests.clear()
        self.ensure_current()

    @staticmethod
    def _entry_state(e):
        """(addresses, versions) of an entry's tensors, or None if its weight is gone."""
        w, g, v = e['w'](), e['gamma'](), e['var']()
        if w is None:
            return None
        return ((w.data_ptr(), None if g is None else g.data_ptr(), None if v is None else v.data_ptr()),
                (w._version, None if g is None else g._version, None if v is None else v._version))

    def ensure_current(self):
        """Re-pack if any planned weight (or BatchNorm scale) changed since the last packing.  Host cost: one tuple of version
        counters per entry; device cost: two launches after an optimiser step, none otherwise."""
        if not self.entries:
            return
        from . import ops
        states = [self._entry_state(e) for e in self.entries]
        if any(st is None for st in states):
            self.entries = [e for e, st in zip(self.entries, states) if st is not None]
            states = [st for st in states if st is not None]
            self.dirty = True
            if not self.entries:
                return
        addresses = tuple(st[0] for st in states)
        key = (ops.WEIGHT_EPOCH[0], addresses, tuple(st[1] for st in states))
        if key == self.key and not self.dirty:
            return
        lib = _lib.load()
        n = len(self.entries)
        upload = int(self.dirty or self.items is None or self.key is None or addresses != self.key[1])
        if upload:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('xconv pack plan: the item table changed during a graph capture (call extend() before it)')
            items = (_lib.XPackItem * n)()
            for i, e in enumerate(self.entries):
                Cout, Cin, KS = e['shape']
                wp, gp, vp = addresses[i]
                items[i] = _lib.XPackItem(wp, e['packed'].data_ptr(), gp, vp, e['eps'], Cout, Cin, KS, e['groups'],
                                          int(e['kind'] != 'F'))
            self.items = items
            self.table = torch.empty(lib.dvd_xconv_pack_table_bytes(n), device=self.entries[0]['packed'].device,
                                     dtype=torch.uint8)
        _lib.check(lib.dvd_xconv_pack_many(ctypes.cast(self.items, ctypes.c_void_p), n, _p(self.table),
                                           ctypes.c_size_t(self.table.numel()), upload, _stream()), 'dvd_xconv_pack_many')
        self.key, self.dirty = key, False
        self.launches += 1

    def lookup(self, weight, kind, gamma=None, var=None, eps=0.0):
        planned = getattr(weight, '_dvd_plan', None)
        e = planned.get(kind) if planned is not None else None
        if e is None or e['w']() is not weight:
            return None
        if kind == 'Ts' and not (e['gamma']() is gamma and e['var']() is var and e['eps'] == float(eps)):
            return None
        return e['packed']


PACK_PLAN = _PackPlan()


def xconv_packed_scaled(weight, groups, gamma, var, eps):
    """Transposed packing with row co scaled by gamma[co] / sqrt(var[co] + eps): backward-data through a fused BatchNorm.
    Cached on the weight like xconv_packed; the key also follows gamma (optimiser steps) and the running variance."""
    from . import ops
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    Cout, Cin, KS, _ = w.shape
    Cin *= groups
    lib = _lib.load()
    nbytes = lib.dvd_xconv_packed_bytes(Cout, Cin, KS, groups, 1)
    capturing = torch.cuda.is_current_stream_capturing()
    if not AB['no_packplan']:
        if capturing:
            planned = PACK_PLAN.lookup(weight, 'Ts', gamma, var, eps)
            if planned is not None:
                return planned
        else:
            PACK_PLAN.request(weight, 'Ts', groups, gamma, var, eps)
    key = (w.data_ptr(), weight._version, ops.WEIGHT_EPOCH[0], tuple(w.shape),
           None if gamma is None else (gamma.data_ptr(), gamma._version), var.data_ptr(), var._version, float(eps))
    cache = getattr(weight, '_dvd_xpack', None)
    if cache is None and not capturing:
        cache = {}
        weight._dvd_xpack = cache
    hit = cache.get('Ts') if cache is not None else None
    if not capturing and hit is not None and hit[0] == key:
        return hit[1]
    packed = hit[1] if (not capturing and hit is not None and hit[1].numel() == nbytes) else \
        torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    _lib.check(lib.dvd_xconv_pack_scaled(_p(w), _p(packed), Cout, Cin, KS, groups, 1, _p(gamma), _p(var), float(eps), _stream()),
               'dvd_xconv_pack_scaled')
    if not capturing:
        cache['Ts'] = (key, packed)
    return packed


def xconv_packed(weight, transposed, groups=1):
    """Fragment-ordered, pre-split copy of a conv weight [Cout,Cin/groups,k,k].  The copy hangs on the weight
    tensor OBJECT (not on its address: allocators reuse addresses) and is rebuilt when the weight changed:
    new storage, autograd version counter, or an optimiser step of the fused Adam (ops.WEIGHT_EPOCH)."""
    from . import ops
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    Cout, Cin, KS, _ = w.shape
    Cin *= groups
    lib = _lib.load()
    if not AB['no_packplan']:
        if torch.cuda.is_current_stream_capturing():
            planned = PACK_PLAN.lookup(weight, 'T' if transposed else 'F')
            if planned is not None:         # kept current by the graph's owner (PACK_PLAN.ensure_current before a replay)
                return planned
        else:
            PACK_PLAN.request(weight, 'T' if transposed else 'F', groups)
    if torch.cuda.is_current_stream_capturing():
        # HIP-graph capture without a plan entry: the pack launch must be PART of the graph (a replay runs no Python, and
        # the weights change between replays), into a buffer of the graph's own pool -- never served from the cache
        nbytes = lib.dvd_xconv_packed_bytes(Cout, Cin, KS, groups, int(transposed))
        packed = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
        _lib.check(lib.dvd_xconv_pack(_p(w), _p(packed), Cout, Cin, KS, groups, int(transposed), _stream()), 'dvd_xconv_pack')
        return packed
    key = (w.data_ptr(), weight._version, ops.WEIGHT_EPOCH[0], tuple(w.shape))
    cache = getattr(weight, '_dvd_xpack', None)
    if cache is None:
        cache = {}
        weight._dvd_xpack = cache
    hit = cache.get(bool(transposed))
    if hit is not None and hit[0] == key:
        return hit[1]
    nbytes = lib.dvd_xconv_packed_bytes(Cout, Cin, KS, groups, int(transposed))
    if nbytes == 0:
        raise RuntimeError('xconv: unsupported weight shape %s' % (tuple(w.shape),))
    packed = hit[1] if (hit is not None and hit[1].numel() == nbytes) else torch.empty(nbytes, device=w.device,
                                                                                       dtype=torch.uint8)
    _lib.check(lib.dvd_xconv_pack(_p(w), _p(packed), Cout, Cin, KS, groups, int(transposed), _stream()), 'dvd_xconv_pack')
    cache[bool(transposed)] = (key, packed)
    return packed


def _xconv_run(x, packed, Cout, KS, bias=None, residual=None, mask_src=None, relu_in=False, relu_out=False,
               res_relu=False, groups=1, bn=None, x_amax=None, y_amax=None, stride=1, out_hw=None):
    """bn = (gamma | None, beta | None, mean, var, eps): eval-mode BatchNorm of the output, fused into the epilogue.
    x_amax: the input's max|x| scalar (computed here if not given); y_amax: optional zeroed 1-element tensor that
    receives max|y|.  stride 2: the 3x3 / stride 2 / padding 1 forward; stride -2: its backward-data pass (x is the gradient
    of the strided output, out_hw the full-resolution height and width), see include/dvd_hip.h."""
    N, Cin, H, W = x.shape
    h16 = _is16(x)
    if x_amax is None and not h16:
        x_amax = amax_of(x)
    flags = int(bool(relu_in)) | (int(bool(relu_out)) << 1) | (int(bool(res_relu)) << 2)
    if stride == 2:
        y = torch.empty(N, Cout, (H + 1) // 2, (W + 1) // 2, device=x.device, dtype=x.dtype)
        flags |= 8
    elif stride == -2:
        if ((out_hw[0] + 1) // 2, (out_hw[1] + 1) // 2) != (H, W):
            raise RuntimeError('xconv: a %dx%d gradient is not the stride-2 image of %dx%d' % (H, W, out_hw[0], out_hw[1]))
        H, W = int(out_hw[0]), int(out_hw[1])
        y = torch.empty(N, Cout, H, W, device=x.device, dtype=x.dtype)
        flags |= 16
    else:
        y = torch.empty(N, Cout, H, W, device=x.device, dtype=x.dtype)
    lib = _lib.load()
    bnp = None
    if bn is not None:
        g, b, m, v, eps = bn
        bnp = ctypes.byref(_lib.BnParams(g.data_ptr() if g is not None else None, b.data_ptr() if b is not None else None,
                                         m.data_ptr(), v.data_ptr(), float(eps)))
    if h16:        # fp16 activations: no operand scale, two MFMAs per product (csrc/xconv.hip IN16 / OUT16)
        for t, name in ((residual, 'residual'), (mask_src, 'mask source')):
            if t is not None and not _is16(t):
                raise RuntimeError('xconv: fp16 convolution with an fp32 %s' % name)
        _lib.check(lib.dvd_xconv_fwd_h(_p(x), _p(packed), _p(bias), _p(residual), _p(mask_src), bnp, _p(y), _p(y_amax), N, Cin,
                                       Cout, H, W, KS, groups, flags, 1, _stream()), 'dvd_xconv_fwd_h')
        return y
    _lib.check(lib.dvd_xconv_fwd(_p(x), _p(x_amax), _p(packed), _p(bias), _p(residual), _p(mask_src), bnp, _p(y), _p(y_amax),
                                 N, Cin, Cout, H, W, KS, groups, flags, _stream()), 'dvd_xconv_fwd')
    return y


STATS = {'sites_premasked': 0, 'sites_masked': 0, 'sites_no_pass': 0}      # how many BatchNorm+ReLU sites took which backward (tests read it)


class _Site(object):
    """Hand-over between a BatchNorm+ReLU site and the convolution that consumes its output (round 3).  The site's backward
    needs g = gy * [y > 0]; the consumer's backward-data kernel PRODUCES gy and has y in hand (its own saved input), so it
    applies the mask in its epilogue (`mask_src`) and records which tensor it wrote.  If the gradient the site receives is
    exactly that tensor, unmodified (same storage address, same version counter: autograd neither replaced it by a sum nor
    accumulated into it), the site's mask pass shrinks to the per-channel sums (4 instead of 12 bytes per element).  Masking
    is idempotent, so a consumer that masks although the site will mask again (several consumers) costs time, never
    correctness; the proof obligation sits with the site alone."""
    __slots__ = ('ref', 'version', 'amax')

    def __init__(self):
        self.ref, self.version, self.amax = None, -1, None

    def wrote(self, g, amax=None):
        """amax: the device scalar max|g| the writing kernel's epilogue produced (the site's operand scale).
        The hand-over is by tensor OBJECT (a weak reference), not by address: a sum (gA + gB) + gC that the allocator
        happens to place at a freed gA's address has version 0 like every kernel-written tensor, but it is another object."""
        import weakref
        self.ref, self.version, self.amax = weakref.ref(g), g._version, amax

    def is_exactly(self, g):
        return self.ref is not None and self.ref() is g and self.version == g._version


class _XConv(torch.autograd.Function):
    """y = conv2d(act(x), w, stride 1, padding k//2) + bias + res'   with act = ReLU or identity and
    res' = residual or relu(residual).  Forward and backward-data on csrc/xconv.hip; backward-weight on
    csrc/xwgrad.hip (deterministic).

    Fused gradient joins (round 3).  `alias`: the Function also returns its input as a second differentiable output; the
    OTHER consumers of x (a residual connection, a shortcut convolution) take that alias instead of x, so their gradient
    arrives HERE, as an argument of backward, and is added in the backward-data kernel's epilogue (`residual` operand) --
    autograd's own accumulation (an ATen add over the whole tensor: read 8, write 4 bytes per element) never runs.  With
    relu_in the epilogue masks AFTER the add, (dgrad + g_alias) * [x > 0], so the alias's consumer must hand over its
    gradient UNMASKED: that is `res_unmasked` on the consuming convolution, whose residual is relu(alias) (the
    ResidualConvUnit of the MiDaS decoder: both terms carry the same mask [x > 0])."""

    @staticmethod
    def forward(ctx, x, x_amax, weight, bias, residual, relu_in, res_relu, groups=1, alias=False, res_unmasked=False,
                in_site=None):
        ctx.set_materialize_grads(False)
        ctx.in_site = in_site if (in_site is not None and not relu_in and not AB['no_maskfuse']) else None
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        Cout, _, KS, _ = weight.shape
        # fp16 activations carry no operand scale; their epilogue folds max|y| into the overflow guard's forward monitor
        y_amax = None if _is16(x) else new_scalar(x.device)
        y = _xconv_run(x, xconv_packed(weight, False, groups), Cout, KS, bias=bias, residual=residual, relu_in=relu_in,
                       res_relu=res_relu, groups=groups, x_amax=x_amax, y_amax=_fwd_monitor() if _is16(x) else y_amax)
        ctx.save_for_backward(x, residual if (res_relu and not res_unmasked) else None, x_amax)
        ctx.wparam = weight          # the tensor object that carries the packed copies
        ctx.cfg = (bool(relu_in), bool(res_relu), bias is not None, residual is not None, groups, bool(res_unmasked))
        if y_amax is not None:
            ctx.mark_non_differentiable(y_amax)
        if alias:
            return y, y_amax, x
        return y, y_amax

    @staticmethod
    def backward(ctx, gy, _g_amax, g_alias=None):
        x, residual, x_amax = ctx.saved_tensors
        weight = ctx.wparam
        relu_in, res_relu, has_bias, has_res, groups, res_unmasked = ctx.cfg
        if gy is None:                      # only the alias was used downstream
            return g_alias, None, None, None, None, None, None, None, None, None, None
        gy = gy.contiguous()
        Cout, Cin, KS, _ = weight.shape
        need = ctx.needs_input_grad
        gx = gw = gb = gr = None
        h16 = _is16(gy)
        want_amax = (need[0] or need[2]) and not h16
        if has_bias and need[3] and gy.is_cuda and gy.dtype == torch.float32 and gy.dim() == 4 and not AB['no_chansum']:
            # the bias gradient and (if nobody attached it) max|gy| from ONE read of gy (rounds 1-5: ATen sum + dvd_amax)
            gb = chansum(gy, want_amax=want_amax and known_amax(gy) is None)
        g_amax = amax_of(gy) if want_amax else None   # one reduction, shared by both gradient kernels
        if need[0]:
            gx_amax = _gs(3) if h16 else new_scalar(gy.device)      # fp16: the loss-scale policy's observed maximum
            gx = _xconv_run(gy, xconv_packed(weight, True, groups), Cin * groups, KS,
                            mask_src=x if (relu_in or ctx.in_site is not None) else None,
                            groups=groups, x_amax=g_amax, y_amax=gx_amax,
                            residual=g_alias.contiguous() if g_alias is not None else None)
            if not h16:
                set_amax(gx, gx_amax)     # (used by the next backward if autograd hands this very tensor on)
            if ctx.in_site is not None:
                ctx.in_site.wrote(gx, None if h16 else gx_amax)     # x is a BatchNorm+ReLU site's output: [x > 0] is already applied
        if need[2]:
            gw = xconv_wgrad(x, gy, weight.shape, relu_in, groups, x_amax=x_amax, g_amax=g_amax)
        if has_bias and need[3] and gb is None:
            gb = gy.sum((0, 2, 3), dtype=torch.float32) * _gs(1) if h16 else gy.sum((0, 2, 3))
        if has_res and need[4]:
            gr = gy * (residual > 0).to(gy.dtype) if (res_relu and not res_unmasked) else gy
        return gx, None, gw, gb, gr, None, None, None, None, None, None


class _XConvS2(torch.autograd.Function):
    """y = conv2d(x, w, bias, stride 2, padding 1) for a 3x3 kernel, dense or grouped with >= 32 channels per group
    (torchvision's Bottleneck.conv2 at the entry of ResNeXt stages 2-4, third_party/midas_blocks.py:35-50).  Rounds 2-5 ran
    the stride-1 kernel and sub-sampled its output (4x the products, a full-resolution round trip); now the forward stages the
    haloed input tile as four phase planes (csrc/xconv.hip, XArgs::S2) and the backward-data kernel reads the compact
    gradient as if it were zero-interleaved (XArgs::ZI).  The weight gradient still takes the interleaved gradient from
    csrc/pool.hip (dvd_subsample2_bwd) into the stride-1 kernels of csrc/xwgrad3.hip."""

    @staticmethod
    def forward(ctx, x, x_amax, weight, bias, groups, in_site):
        ctx.set_materialize_grads(False)
        ctx.in_site = in_site if (in_site is not None and not AB['no_maskfuse']) else None
        x = x.contiguous()
        Cout = weight.shape[0]
        y_amax = None if _is16(x) else new_scalar(x.device)
        y = _xconv_run(x, xconv_packed(weight, False, groups), Cout, 3, bias=bias, groups=groups, x_amax=x_amax,
                       y_amax=_fwd_monitor() if _is16(x) else y_amax, stride=2)
        ctx.save_for_backward(x, x_amax)
        ctx.wparam = weight
        ctx.cfg = (bias is not None, groups)
        if y_amax is not None:
            ctx.mark_non_differentiable(y_amax)
        return y, y_amax

    @staticmethod
    def backward(ctx, gy, _g_amax):
        x, x_amax = ctx.saved_tensors
        weight = ctx.wparam
        has_bias, groups = ctx.cfg
        if gy is None:
            return None, None, None, None, None, None
        gy = gy.contiguous()
        need = ctx.needs_input_grad
        N, Cin, H, W = x.shape
        h16 = _is16(gy)
        gx = gw = gb = None
        want_amax = (need[0] or need[2]) and not h16
        if has_bias and need[3] and gy.dtype == torch.float32 and not AB['no_chansum']:
            gb = chansum(gy, want_amax=want_amax and known_amax(gy) is None)
        g_amax = amax_of(gy) if want_amax else None
        if need[0]:
            gx_amax = _gs(3) if h16 else new_scalar(gy.device)
            gx = _xconv_run(gy, xconv_packed(weight, True, groups), Cin, 3, mask_src=x if ctx.in_site is not None else None,
                            groups=groups, x_amax=g_amax, y_amax=gx_amax, stride=-2, out_hw=(H, W))
            if not h16:
                set_amax(gx, gx_amax)
            if ctx.in_site is not None:
                ctx.in_site.wrote(gx, None if h16 else gx_amax)
        if need[2]:
            gyf = torch.empty(N, gy.shape[1], H, W, device=gy.device, dtype=gy.dtype)       # zero-interleaved gradient
            _lib.check(_lib.load().dvd_subsample2_bwd(_p(gy), _p(gyf), int(h16), N * gy.shape[1], H, W, _stream()),
                       'dvd_subsample2_bwd')
            gw = xconv_wgrad(x, gyf, weight.shape, False, groups, x_amax=x_amax, g_amax=g_amax)
        if has_bias and need[3] and gb is None:
            gb = gy.sum((0, 2, 3), dtype=torch.float32) * _gs(1) if h16 else gy.sum((0, 2, 3))
        return gx, None, gw, gb, None, None


def xconv_s2_supported(x, weight, groups):
    """3x3 / stride 2 / padding 1 on the strided forms of csrc/xconv.hip: whole 16-channel chunks per group (the
    buffer-addressed main loop), grouped layers with at least 32 channels per group."""
    cpg = weight.shape[1]
    return (x.is_cuda and x.dtype in ACT_DTYPES and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3) and
            cpg % 16 == 0 and (weight.shape[0] // groups) % 16 == 0 and (groups == 1 or cpg >= 32) and
            not AB['no_xconv'] and not AB['no_s2'])


def _xconv_s2(x, weight, bias, groups=1):
    x_amax = None if _is16(x) else amax_of(x)
    y, y_amax = _XConvS2.apply(x, x_amax, weight, bias, groups, getattr(x, '_dvd_site', None))
    return set_amax(y, y_amax)


def _xconv(x, weight, bias, residual, relu_in, res_relu, groups=1, alias=False, res_unmasked=False):
    """_XConv with the max|.| scalars threaded through: the input's is looked up (or computed), the output's attached.
    alias=True returns (y, alias of x), see _XConv."""
    x_amax = None if _is16(x) else amax_of(x)
    in_site = getattr(x, '_dvd_site', None)
    if alias:
        y, y_amax, xa = _XConv.apply(x, x_amax, weight, bias, residual, relu_in, res_relu, groups, True, res_unmasked, in_site)
        return set_amax(y, y_amax), set_amax(xa, x_amax)
    y, y_amax = _XConv.apply(x, x_amax, weight, bias, residual, relu_in, res_relu, groups, False, res_unmasked, in_site)
    return set_amax(y, y_amax)


def wgrad_reports_rowsum(wshape, groups):
    """Does xconv_wgrad(..., rowsum=t) fill t?  (dense 1x1: csrc/xwgrad3.hip dvd_xwgrad1s_rowsum)"""
    return wshape[2] == 1 and groups == 1 and not AB['no_xwgrad3']


def xconv_wgrad(x, gy, wshape, relu_in, groups=1, x_amax=None, g_amax=None, rowsum=None):
    """dW[co][ci][tap] = sum_{n,p} gy[n][co][p] * act(x)[n][ci][p + tap]."""
    lib = _lib.load()
    if _is16(gy):        # fp16 operands: one MFMA per product, result times 1 / (loss scale) (csrc/xwgrad3.hip H16)
        if not _is16(x):
            raise RuntimeError('xconv_wgrad: fp16 gradient with an fp32 activation')
        if wshape[2] not in (1, 3) or (groups > 1 and wshape[2] != 3):
            raise RuntimeError('xconv: the fp16 weight gradient exists for 1x1 and 3x3 kernels only (got %s)' % (tuple(wshape),))
        N, Cin, H, W = x.shape
        gw = torch.empty(wshape, device=x.device, dtype=torch.float32)
        if wshape[2] == 3:
            ws = _workspace(lib.dvd_xwgrad3_workspace_bytes(N, Cin, wshape[0], H, W, groups), x.device)
            _lib.check(lib.dvd_xwgrad3_h(_p(x), _p(gy), _p(_gs(1)), _p(gw), _p(ws), ctypes.c_size_t(ws.numel()), N, Cin, wshape[0],
                                         H, W, groups, int(bool(relu_in)), _stream()), 'dvd_xwgrad3_h')
        else:
            ws = _workspace(lib.dvd_xwgrad1s_workspace_bytes(N, Cin, wshape[0], H, W), x.device)
            _lib.check(lib.dvd_xwgrad1s_h(_p(x), _p(gy), _p(_gs(1)), _p(gw), _p(ws), ctypes.c_size_t(ws.numel()), N, Cin, wshape[0],
                                          H, W, int(bool(relu_in)), _stream()), 'dvd_xwgrad1s_h')
        return gw
    if x_amax is None:
        x_amax = amax_of(x)
    if g_amax is None:
        g_amax = amax_of(gy)
    if groups > 1:                                                          # grouped: 3x3 only (ResNeXt stage 4)
        if wshape[2] != 3:
            raise RuntimeError('xconv: grouped weight gradient exists for 3x3 kernels only')
        N, Cin, H, W = x.shape
        gw = torch.empty(wshape, device=x.device, dtype=torch.float32)
        ws = _workspace(lib.dvd_xwgrad3_workspace_bytes(N, Cin, wshape[0], H, W, groups), x.device)
        _lib.check(lib.dvd_xwgrad3(_p(x), _p(x_amax), _p(gy), _p(g_amax), _p(gw), _p(ws), ctypes.c_size_t(ws.numel()), N, Cin,
                                   wshape[0], H, W, groups, int(bool(relu_in)), _stream()), 'dvd_xwgrad3')
        return gw
    if wshape[2] in (1, 3) and not AB['no_xwgrad3']:      # split-operand MFMA (csrc/xwgrad3.hip)
        N, Cin, H, W = x.shape
        Cout = wshape[0]
        gw = torch.empty(wshape, device=x.device, dtype=torch.float32)
        ws_bytes = (lib.dvd_xwgrad3_workspace_bytes(N, Cin, Cout, H, W, 1) if wshape[2] == 3 else
                    lib.dvd_xwgrad1s_workspace_bytes(N, Cin, Cout, H, W))
        ws = _workspace(ws_bytes, x.device)
        if wshape[2] == 3:
            _lib.check(lib.dvd_xwgrad3(_p(x), _p(x_amax), _p(gy), _p(g_amax), _p(gw), _p(ws), ctypes.c_size_t(ws.numel()), N,
                                       Cin, Cout, H, W, 1, int(bool(relu_in)), _stream()), 'dvd_xwgrad3')
        else:
            # rowsum: [Cout] tensor that receives sum_{n,p} gy[n][co][p] (the wide kernel sums the rows it stages anyway)
            _lib.check(lib.dvd_xwgrad1s_rowsum(_p(x), _p(x_amax), _p(gy), _p(g_amax), _p(gw), _p(rowsum), _p(ws),
                                               ctypes.c_size_t(ws.numel()), N, Cin, Cout, H, W, int(bool(relu_in)), _stream()),
                       'dvd_xwgrad1s_rowsum')
        return gw
    if wshape[2] in (5, 7, 11) and not AB['no_xwgrad3']:      # split-operand MFMA with KS kernel rows (csrc/xwgrad3.hip xwgradk)
        N, Cin, H, W = x.shape
        Cout, _, KS, _ = wshape
        gw = torch.empty(wshape, device=x.device, dtype=torch.float32)
        ws = _workspace(lib.dvd_xwgradk_workspace_bytes(N, Cin, Cout, H, W, KS), x.device)
        _lib.check(lib.dvd_xwgradk(_p(x), _p(x_amax), _p(gy), _p(g_amax), _p(gw), _p(ws), ctypes.c_size_t(ws.numel()), N, Cin, Cout,
                                   H, W, KS, int(bool(relu_in)), _stream()), 'dvd_xwgradk')
        return gw
    if (wshape[2] in (1, 3) and not AB['no_xwgrad']) or wshape[2] in (5, 7, 11):
        # exact-fp32 MFMA kernel (csrc/xwgrad.hip): the A/B fall-back of the small kernels, and -- round 4 -- THE weight gradient
        # of the hourglass's 5x5 / 7x7 / 11x11 inception branches and of the stem's 5x5 space-to-depth form (round 3: MIOpen)
        N, Cin, H, W = x.shape
        Cout, _, KS, _ = wshape
        gw = torch.empty(wshape, device=x.device, dtype=torch.float32)
        nws = lib.dvd_xwgrad_workspace_bytes(N, Cin, Cout, H, W, KS)
        ws = _workspace(nws, x.device)
        _lib.check(lib.dvd_xwgrad(_p(x), _p(gy), _p(gw), _p(ws), ctypes.c_size_t(ws.numel()), N, Cin, Cout, H, W, KS,
                                  int(bool(relu_in)), _stream()), 'dvd_xwgrad')
        return gw
    if AB['no_xwgrad'] and wshape[2] in (1, 3):            # A/B only: MIOpen's weight-gradient kernels
        xin = torch.relu(x) if relu_in else x
        KS = wshape[2]
        return torch.ops.aten.convolution_backward(gy, xin, torch.empty(wshape, device=x.device), None, [1, 1],
                                                   [KS // 2, KS // 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    raise RuntimeError('xconv: no weight-gradient kernel for a %dx%d convolution (1, 3, 5, 7, 11 are covered)' % (wshape[2], wshape[3]))


class _XConvBn(torch.autograd.Function):
    """y = act(bn_eval(conv2d(x, w) + cbias) (+ residual)): the convolution kernel's epilogue applies the BatchNorm, so
    the pre-BN tensor is never written.  Backward: one pass masks the output gradient (g = gy * [y > 0]) and sums it per
    channel (dbeta); backward-data runs on g with the weights packed transposed AND scaled by gamma * rstd; backward-weight
    runs on g unscaled, and a tiny kernel derives dW, dgamma and the conv-bias gradient from it (csrc/bnrelu.hip)."""

    @staticmethod
    def forward(ctx, x, x_amax, weight, cbias, gamma, beta, mean, var, eps, residual, relu, groups, alias=False, in_site=None,
                out_site=None):
        ctx.set_materialize_grads(False)    # (alias: see _XConv -- the gradient of the input's other consumers arrives in backward)
        ctx.in_site = in_site if (in_site is not None and not AB['no_maskfuse']) else None      # see _Site
        ctx.out_site = out_site if relu else None
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        Cout, _, KS, _ = weight.shape
        y_amax = None if _is16(x) else new_scalar(x.device)
        y = _xconv_run(x, xconv_packed(weight, False, groups), Cout, KS, bias=cbias, residual=residual, relu_out=relu,
                       groups=groups, bn=(gamma, beta, mean, var, eps), x_amax=x_amax,
                       y_amax=_fwd_monitor() if _is16(x) else y_amax)
        ctx.save_for_backward(x, y if relu else None, gamma, mean, var, cbias, x_amax)
        ctx.wparam = weight
        ctx.cfg = (float(eps), bool(relu), residual is not None, groups)
        if y_amax is not None:
            ctx.mark_non_differentiable(y_amax)
        if alias:
            return y, y_amax, x
        return y, y_amax

    @staticmethod
    def backward(ctx, gy, _g_amax, g_alias=None):
        x, y, gamma, mean, var, cbias, x_amax = ctx.saved_tensors
        weight = ctx.wparam
        eps, relu, has_res, groups = ctx.cfg
        if gy is None:                      # only the alias was used downstream
            return (g_alias,) + (None,) * 14
        gy = gy.contiguous()
        Cout, Cing, KS, _ = weight.shape
        N, _, H, W = gy.shape
        need = ctx.needs_input_grad
        lib = _lib.load()
        # masked gradient + per-channel sums
        dbeta = torch.empty(Cout, device=gy.device, dtype=torch.float32)
        # the consumer's backward-data epilogue has applied [y > 0] already and nothing touched the tensor since: sums only
        premasked = relu and ctx.out_site is not None and ctx.out_site.is_exactly(gy) and not AB['no_maskfuse']
        mask = relu and not premasked
        need_w = need[2] or (gamma is not None and need[4]) or (cbias is not None and need[3])
        # ... and if the weight gradient runs on the 1x1 kernel, that kernel reports the per-channel sums of the rows it stages
        # and the consumer's epilogue has left max|g|: no pass of this site's own at all
        no_pass = (premasked and need_w and ctx.out_site.amax is not None and wgrad_reports_rowsum(weight.shape, groups) and
                   AB['rowsum'])
        if relu:
            STATS['sites_no_pass' if no_pass else ('sites_premasked' if premasked else 'sites_masked')] += 1
        g = torch.empty_like(gy) if mask else gy
        h16 = _is16(gy)
        if no_pass:
            g_amax = ctx.out_site.amax
        else:
            ws = _workspace(lib.dvd_bnrelu_bwd_workspace_bytes(N, Cout, H * W), gy.device)
            # max|masked gradient|, folded in by the mask pass (it reads every element anyway); fp16: the policy's observed maximum
            g_amax = _gs(3) if h16 else new_scalar(gy.device)
            _lib.check(lib.dvd_bnrelu_bwd_t(_p(gy), _p(y) if mask else None, None, _p(var), _p(mean), _p(var), eps, None,
                                            _p(g) if mask else None, None, _p(dbeta), _p(ws), ctypes.c_size_t(ws.numel()),
                                            int(h16), _p(_gs(1)) if h16 else None, N, Cout, H * W, int(mask), _p(g_amax),
                                            _stream()), 'dvd_bnrelu_bwd')
        gx = gw = gcb = gg = None
        if need[0]:
            gx_amax = _gs(3) if h16 else new_scalar(gy.device)
            gx = _xconv_run(g, xconv_packed_scaled(weight, groups, gamma, var, eps), Cing * groups, KS, groups=groups,
                            x_amax=g_amax, y_amax=gx_amax,
                            residual=g_alias.contiguous() if g_alias is not None else None,   # + the other consumers' gradient
                            mask_src=x if ctx.in_site is not None else None)                  # ... * [x > 0] for the site x came from
            if not h16:
                set_amax(gx, gx_amax)
            if ctx.in_site is not None:
                ctx.in_site.wrote(gx, None if h16 else gx_amax)
        elif g_alias is not None:
            gx = g_alias
        if need_w:
            gw = xconv_wgrad(x, g, weight.shape, False, groups, x_amax=x_amax, g_amax=g_amax, rowsum=dbeta if no_pass else None)
            gg = torch.empty_like(gamma) if gamma is not None else None
            gcb = torch.empty_like(cbias) if cbias is not None else None
            _lib.check(lib.dvd_convbn_finalize(_p(weight.detach()), _p(gw), _p(dbeta), _p(gamma), _p(mean), _p(var), eps,
                                               _p(cbias), Cout, Cing * KS * KS, _p(gg), _p(gcb), _stream()),
                       'dvd_convbn_finalize')
        return (gx, None, gw, gcb, gg, (dbeta if need[5] else None), None, None, None, (g if has_res else None), None, None, None,
                None, None)


def conv_bn_act(conv, bn, x, residual=None, relu=True, alias=False):
    """relu(bn(conv(x)) (+ residual)) for an nn.Conv2d followed by an eval-mode nn.BatchNorm2d.  Convolutions the xconv
    kernels cover run as ONE launch (BatchNorm, residual and ReLU in the epilogue); everything else (CPU tensors,
    training-mode statistics, the 8/16-per-group and strided 3x3 convolutions) is conv(x) followed by the fused
    BatchNorm+ReLU kernel / the ATen ops.
    alias=True returns (y, x'): x' carries x's values and must be used by every OTHER consumer of x (the block's shortcut);
    on the fused path their gradient is then added inside this convolution's backward-data kernel instead of by autograd's
    accumulation pass (see _XConv); on the other paths x' is x itself."""
    if (x.is_cuda and x.dtype in ACT_DTYPES and not bn.training and bn.track_running_stats and
            isinstance(conv, nn.Conv2d) and not AB['no_bnfuse']):
        xin = None
        if xconv_supported(conv, x):
            xin = x
        elif (conv.kernel_size == (1, 1) and conv.groups == 1 and tuple(conv.padding) == (0, 0) and
              conv.stride[0] == conv.stride[1] and conv.stride[0] > 1 and conv.weight.dtype == torch.float32 and
              not AB['no_xconv']):
            xin = _subsample(x, conv.stride[0])
        if xin is not None:
            gamma, beta = (bn.weight, bn.bias) if bn.affine else (None, None)
            x_amax = None if _is16(xin) else amax_of(xin)
            in_site = getattr(xin, '_dvd_site', None)           # xin is a BatchNorm+ReLU site's output (see _Site)
            out_site = _Site() if relu else None
            if alias and xin is x and not AB['no_alias']:
                y, y_amax, xa = _XConvBn.apply(xin, x_amax, conv.weight, conv.bias, gamma, beta, bn.running_mean,
                                               bn.running_var, bn.eps, residual, relu, conv.groups, True, in_site, out_site)
                y._dvd_site = out_site
                return set_amax(y, y_amax), set_amax(xa, x_amax)
            y, y_amax = _XConvBn.apply(xin, x_amax, conv.weight, conv.bias, gamma, beta, bn.running_mean,
                                       bn.running_var, bn.eps, residual, relu, conv.groups, False, in_site, out_site)
            y._dvd_site = out_site
            return (set_amax(y, y_amax), x) if alias else set_amax(y, y_amax)
    y = bn_eval_relu(bn, conv(x), residual=residual, relu=relu)
    return (y, x) if alias else y


def xconv_supported(conv, x):
    k = conv.kernel_size
    return (x.is_cuda and x.dtype in ACT_DTYPES and conv.weight.dtype == torch.float32 and
            (conv.groups == 1 or (k[0] == 3 and conv.in_channels // conv.groups >= 32)) and
            k[0] == k[1] and k[0] % 2 == 1 and k[0] <= 11 and tuple(conv.stride) == (1, 1) and
            tuple(conv.dilation) == (1, 1) and tuple(conv.padding) == (k[0] // 2, k[0] // 2) and
            conv.padding_mode == 'zeros' and not AB['no_xconv'])


def xconv2d(conv, x, relu_in=False, residual=None, res_relu=False, alias=False, res_unmasked=False):
    """`conv(relu(x) if relu_in else x) + (relu(residual) if res_relu else residual)` for an nn.Conv2d `conv`.
    GPU fp32 tensors of a dense stride-1 'same' convolution run on the HIP kernels; CPU tensors (the oracle /
    golden-fixture generator instantiates these modules on the CPU) take the ATen ops the reference uses."""
    if xconv_supported(conv, x):
        if AB['no_alias']:
            y = _xconv(x, conv.weight, conv.bias, residual, relu_in, res_relu, conv.groups)
            return (y, x) if alias else y
        return _xconv(x, conv.weight, conv.bias, residual, relu_in, res_relu, conv.groups, alias, res_unmasked)
    if x.is_cuda and x.dtype in ACT_DTYPES and conv.groups == 1 and tuple(conv.stride) == (1, 1) and \
            not AB['no_xconv']:
        raise RuntimeError('xconv2d: convolution %r is not covered by the HIP kernels' % (conv,))
    y = conv(F.relu(x) if relu_in else x)
    if residual is not None:
        y = y + (F.relu(residual) if res_relu else residual)
    return (y, x) if alias else y


class XConv2d(nn.Conv2d):
    """Drop-in nn.Conv2d (same parameters / state_dict keys) whose stride-1 'same' case (dense, or grouped 3x3 with
    at least 32 channels per group) runs on the HIP kernels; a strided 1x1 convolution without padding (the ResNeXt
    down-sampling shortcut) is the 1x1 kernel on the sub-sampled input, a stride-2 3x3 'same' convolution runs on the strided
    forms of the kernel (_XConvS2, round 6; other strides, and DVD_AB=no_s2: the stride-1 kernel's output sub-sampled).
    Anything else, and CPU tensors, take ATen."""

    def forward(self, x):
        if xconv_supported(self, x):
            return _xconv(x, self.weight, self.bias, None, False, False, self.groups)
        k, st = self.kernel_size, self.stride
        if (x.is_cuda and x.dtype in ACT_DTYPES and k == (3, 3) and st[0] == st[1] and st[0] > 1 and
                tuple(self.padding) == (1, 1) and tuple(self.dilation) == (1, 1) and
                (self.groups == 1 or self.in_channels // self.groups >= 32) and not AB['no_xconv']):
            if st[0] == 2 and xconv_s2_supported(x, self.weight, self.groups):
                return _xconv_s2(x, self.weight, self.bias, self.groups)
            # out[i][j] of a stride-s 'same' 3x3 convolution is out1[s*i][s*j] of the stride-1 one
            y = _xconv(x, self.weight, self.bias, None, False, False, self.groups)
            return _subsample(y, st[0])
        if (x.is_cuda and x.dtype in ACT_DTYPES and self.kernel_size == (1, 1) and self.groups == 1 and
                tuple(self.padding) == (0, 0) and self.stride[0] == self.stride[1] and self.stride[0] > 1 and
                not AB['no_xconv']):
            st = self.stride[0]
            return _xconv(_subsample(x, st), self.weight, self.bias, None, False, False)
        return super().forward(x)
