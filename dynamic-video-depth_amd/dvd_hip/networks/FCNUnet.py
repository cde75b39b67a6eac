"""`--use_cnn`: the U-Net scene-flow network on this package's convolution kernels.

Mirrors /root/reference/networks/FCNUnet.py:21-92 (`FCNUnet`) with the blocks of networks/blocks.py:36-102
(`Conv2dBlock` = ReflectionPad2d -> Conv2d(padding=0) -> Identity norm -> LeakyReLU(0.2); `DoubleConv2dBlock` = two of them)
in the configuration the reference Model builds (models/scene_flow_motion_field.py:102-105): conv_setup = {norm none,
activation lrelu, pad reflect, stride 1}, n_down = --n_down, feat 32, double_conv blocks, in_channel 4 (xyz + t) or 3,
out_channel 3.  Same module tree and state_dict keys (down_%02d.model.{0,1}.conv.*, mid_conv.*, up_%04d.*, output_conv.conv.*),
so checkpoints interchange.

GPU fp32 tensors: every convolution runs on csrc/xconv.hip / csrc/xwgrad3.hip (`conv._xconv`: forward, backward-data and
backward-weight).  The kernels pad with zeros, the network pads by REFLECTION, so a block reflects its input explicitly
(ATen reflection_pad2d: an HBM copy) and takes the interior of the zero-padded "same" convolution of the padded tensor -- which
is the valid convolution the reference computes.  AvgPool2d(3, 2, 1) is csrc/pool.hip (round 6: `conv.AvgPool2d`), the x2
bilinear up-sampling (align_corners=True) csrc/upsample.hip; the channel concatenations and LeakyReLU stay ATen elementwise /
copy kernels.  CPU tensors take the ATen
modules, i.e. the reference's own arithmetic (the oracle / golden generator instantiate this class on the CPU).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..conv import AvgPool2d, _xconv, upsample_bilinear2x


class Conv2dBlock(nn.Module):
    def __init__(self, input_dim, output_dim, kernel_size, padding=0, activation='lrelu'):
        super().__init__()
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, 1, padding=0, bias=True)
        self.pad = nn.ReflectionPad2d(padding)
        self.norm = nn.Identity()
        self.activation = nn.LeakyReLU(0.2, inplace=True) if activation == 'lrelu' else nn.Identity()
        self.padding, self.lrelu = int(padding), activation == 'lrelu'

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            p = self.padding
            xp = F.pad(x, (p, p, p, p), mode='reflect') if p else x
            y = _xconv(xp.contiguous(), self.conv.weight, self.conv.bias, None, False, False)
            if p:
                y = y[:, :, p:-p, p:-p].contiguous()
            return F.leaky_relu(y, 0.2) if self.lrelu else y
        return self.activation(self.norm(self.conv(self.pad(x))))


class DoubleConv2dBlock(nn.Module):
    def __init__(self, input_dim, output_dim, kernel_size, padding=0):
        super().__init__()
        self.model = nn.Sequential(Conv2dBlock(input_dim, output_dim, kernel_size, padding),
                                   Conv2dBlock(output_dim, output_dim, kernel_size, padding))

    def forward(self, x):
        return self.model(x)


class FCNUnet(nn.Module):
    def __init__(self, conv_setup=None, n_down=4, feat=32, block_type='double_conv', down_sample_type='avgpool', in_channel=2,
                 out_channel=64, dialated_pool=False, output_activation=None):
        super().__init__()
        cs = conv_setup or {}
        if (block_type != 'double_conv' or down_sample_type != 'avgpool' or output_activation is not None or
                cs.get('norm', 'none') != 'none' or cs.get('activation', 'lrelu') != 'lrelu' or
                cs.get('pad_type', 'reflect') != 'reflect' or cs.get('stride', 1) != 1):
            raise NotImplementedError('FCNUnet: the configuration of models/scene_flow_motion_field.py:102-105 is implemented '
                                      '(double_conv blocks, avgpool, reflect padding, lrelu, no norm)')
        self.down_sample = AvgPool2d(kernel_size=3, stride=2, padding=1)      # (nn.AvgPool2d on csrc/pool.hip for GPU tensors)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.n_down = n_down
        self.down_conv, self.up_conv = [], []
        ch_in, ch_out = in_channel, feat
        for k in range(n_down):
            self.down_conv.append(DoubleConv2dBlock(ch_in, ch_out, 3, padding=1))
            self.add_module('down_%02d' % k, self.down_conv[-1])
            ch_in, ch_out = ch_out, ch_out * 2
        self.mid_conv = DoubleConv2dBlock(ch_in, ch_in, 3, padding=1)
        k = -1
        for k in range(n_down - 1):
            self.up_conv.append(DoubleConv2dBlock(ch_in * 2, ch_in // 2, 3, padding=1))
            self.add_module('up_%04d' % k, self.up_conv[-1])
            ch_in = ch_in // 2
        self.up_conv.append(DoubleConv2dBlock(ch_in * 2, ch_in, 3, padding=1))       # "matching original unet implementation"
        self.add_module('up_%04d' % (k + 1), self.up_conv[-1])
        self.output_conv = Conv2dBlock(ch_in, out_channel, 1, padding=0, activation='none')
        self.final_act = nn.Identity()

    def forward(self, x):
        if x.shape[2] % (1 << self.n_down) or x.shape[3] % (1 << self.n_down):
            raise RuntimeError('FCNUnet: image size %dx%d is not a multiple of 2^n_down (the skip connections would not line up)'
                               % (x.shape[2], x.shape[3]))
        feat = []
        for module in self.down_conv:
            x = module(x)
            feat.append(x)
            x = self.down_sample(x)
        x = self.mid_conv(x)
        for idm, module in enumerate(self.up_conv):
            up_x = upsample_bilinear2x(x, align_corners=True) if x.is_cuda else self.upsample(x)
            x = module(torch.cat([feat[-(idm + 1)], up_x], 1))
        return self.final_act(self.output_conv(x))
