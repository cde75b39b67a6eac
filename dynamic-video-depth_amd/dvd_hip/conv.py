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
from .ops import _p, _stream, _workspace


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
        _lib.check(lib.dvd_bnrelu_fwd(_p(x), _p(residual), _p(gamma), _p(beta), _p(mean), _p(var), float(eps), _p(y), N, C,
                                      HW, int(relu), _stream()), 'dvd_bnrelu_fwd')
        ctx.save_for_backward(x, y if relu else None, gamma, mean, var)
        ctx.cfg = (N, C, HW, float(eps), int(relu), residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
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
        _lib.check(lib.dvd_bnrelu_bwd(_p(gy), _p(y), _p(x), _p(gamma), _p(mean), _p(var), eps, _p(gx), _p(gr), _p(gg),
                                      _p(gb), _p(ws), ctypes.c_size_t(ws.numel()), N, C, HW, relu, _stream()),
                   'dvd_bnrelu_bwd')
        return gx, gr, gg, gb, None, None, None, None


def bn_eval_relu(bn, x, residual=None, relu=True):
    """relu(bn(x) (+ residual)) for an nn.BatchNorm2d in eval mode (running statistics; gamma / beta keep
    their gradients) on the fused HIP kernel; anything else (training-mode BN, CPU, other dtypes) takes the
    ATen ops the reference uses."""
    if (not bn.training and bn.affine and bn.track_running_stats and x.is_cuda and x.dtype == torch.float32):
        return _BnRelu.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw, align_corners):
        x = x.contiguous()
        N, C, H, W = x.shape
        Ho, Wo = out_hw
        y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=x.dtype)
        lib = _lib.load()
        _lib.check(lib.dvd_upsample_bilinear_fwd(_p(x), _p(y), N * C, H, W, Ho, Wo, int(align_corners), _stream()),
                   'dvd_upsample_bilinear_fwd')
        ctx.shape, ctx.align = (N, C, H, W, Ho, Wo), int(align_corners)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W, Ho, Wo = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(N, C, H, W, device=gy.device, dtype=gy.dtype)
        lib = _lib.load()
        _lib.check(lib.dvd_upsample_bilinear_bwd(_p(gy), _p(gx), N * C, H, W, Ho, Wo, ctx.align, _stream()),
                   'dvd_upsample_bilinear_bwd')
        return gx, None, None


def upsample_bilinear2x(x, align_corners):
    """F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=...) of the MiDaS decoder."""
    if x.is_cuda and x.dtype == torch.float32:
        return _UpsampleBilinear.apply(x, (2 * x.shape[2], 2 * x.shape[3]), bool(align_corners))
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=align_corners)


class _GConv3x3C8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()
        w = w.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        lib = _lib.load()
        _lib.check(lib.dvd_gconv3x3_c8_fwd(_p(x), _p(w), _p(y), N, C, H, W, _stream()), 'dvd_gconv3x3_c8_fwd')
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
            _lib.check(lib.dvd_gconv3x3_c8_bwd_data(_p(gy), _p(w), _p(gx), N, C, H, W, _stream()),
                       'dvd_gconv3x3_c8_bwd_data')
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            nws = lib.dvd_gconv3x3_c8_wgrad_workspace_bytes(N, C, H, W)
            ws = _workspace(nws, x.device)
            _lib.check(lib.dvd_gconv3x3_c8_bwd_weight(_p(x), _p(gy), _p(gw), 0, _p(ws), ctypes.c_size_t(ws.numel()),
                                                      N, C, H, W, _stream()), 'dvd_gconv3x3_c8_bwd_weight')
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
        return gconv3x3_c32(x, self.weight)


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

    def __init__(self, channels):
        if channels % 32:
            raise ValueError('GroupedConv3x3C16 needs a multiple of 32 channels')
        super().__init__(channels, channels, 3, stride=1, padding=1, groups=channels // 16, bias=False)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and not _os.environ.get('DVD_NO_C16'):
            return gconv3x3_c32(x, _pair_groups_of_16(self.weight))
        return F.conv2d(x, self.weight, None, 1, 1, 1, self.groups)


def gconv3x3_c8(x, weight):
    """y = conv2d(x, weight, padding=1, groups=C // 8) for weight [C, 8, 3, 3]."""
    if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32:
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
