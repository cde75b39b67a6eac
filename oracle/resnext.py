"""ResNeXt-101 32x8d (the MiDaS encoder) -- TEST INFRASTRUCTURE, an independent restatement.

The reference obtains its encoder with
    torch.hub.load("facebookresearch/WSL-Images", "resnext101_32x8d_wsl")      third_party/midas_blocks.py:48-50
(hub entry, branch unpinned) which instantiates torchvision 0.10's (dependencies/requirements.txt:56)
    torchvision.models.resnet.ResNet(Bottleneck, [3, 4, 23, 3], groups=32, width_per_group=8)
Neither the hub nor torchvision exists in the build container, so this file restates that published
architecture from its description in plain torch.nn, WITHOUT looking at or importing
the product's `dvd_hip.third_party.MiDaS` -- it is the independent side of the structural pin:

  * tests/golden/make_golden.py patches `torch.hub.load` to return `resnext101_32x8d()`; the reference's own
    `_make_resnet_backbone` (midas_blocks.py:35-45) then assembles `pretrained.layer1..4` from its attributes
    conv1 / bn1 / relu / maxpool / layer1..layer4, exactly as it would from torchvision's object;
  * tests/test_resnext_pin_cpu.py compares state_dict keys + shapes and a seeded forward of the product encoder
    against this one.

Architecture, as published (torchvision 0.10 resnet.py):
  stem      conv 7x7, 64, stride 2, padding 3, no bias -> BatchNorm -> ReLU -> max-pool 3x3, stride 2, padding 1
  stage k   blocks = [3, 4, 23, 3][k], planes = 64 * 2^k, the first block of stages 1..3 has stride 2
  block     width = int(planes * (width_per_group / 64)) * groups
            conv 1x1 (in -> width) -> BN -> ReLU
            conv 3x3 (width -> width, groups = 32, THE STRIDE IS HERE, padding 1) -> BN -> ReLU
            conv 1x1 (width -> 4 planes) -> BN
            + identity, or + BN(conv 1x1 (in -> 4 planes, stride)) when the shape changes ("downsample": a Sequential,
              keys downsample.0 / downsample.1)
            -> ReLU
  no conv has a bias; conv weights are kaiming-normal (fan_out, relu); BN weight 1 / bias 0.
Only conv1, bn1, relu, maxpool, layer1..4 are consumed by the reference (midas_blocks.py:35-45); avgpool / fc are
omitted.  Parity status: pinned to the published structure only -- the WSL weights are not available offline.
"""
import torch.nn as nn


class Block(nn.Module):
    """torchvision `Bottleneck`; attribute names are the state_dict keys."""

    def __init__(self, inplanes, planes, stride, downsample, groups, width_per_group):
        super(Block, self).__init__()
        mid = int(planes * (width_per_group / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, mid, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, kernel_size=3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, 4 * planes, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        z = self.relu(self.bn1(self.conv1(x)))
        z = self.relu(self.bn2(self.conv2(z)))
        z = self.bn3(self.conv3(z))
        z += shortcut
        return self.relu(z)


class Encoder(nn.Module):
    def __init__(self, depths=(3, 4, 23, 3), groups=32, width_per_group=8):
        super(Encoder, self).__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for k, nblocks in enumerate(depths):
            setattr(self, 'layer%d' % (k + 1), self._stage(64 << k, nblocks, 1 if k == 0 else 2, groups, width_per_group))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, planes, nblocks, stride, groups, wpg):
        blocks = []
        for i in range(nblocks):
            s = stride if i == 0 else 1
            down = None
            if s != 1 or self.inplanes != 4 * planes:
                down = nn.Sequential(nn.Conv2d(self.inplanes, 4 * planes, kernel_size=1, stride=s, bias=False),
                                     nn.BatchNorm2d(4 * planes))
            blocks.append(Block(self.inplanes, planes, s, down, groups, wpg))
            self.inplanes = 4 * planes
        return nn.Sequential(*blocks)

    def forward(self, x):                       # not used by MiDaS (it takes the stages apart); for tests
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def resnext101_32x8d():
    return Encoder((3, 4, 23, 3), groups=32, width_per_group=8)
