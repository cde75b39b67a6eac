"""Oracle (test-only): the `--use_cnn` U-Net scene-flow network as a function of a weight dict.

Functional restatement of /root/reference/networks/FCNUnet.py:21-92 with the blocks of networks/blocks.py:36-102 in the
configuration of models/scene_flow_motion_field.py:102-105 (double_conv blocks, reflection padding, LeakyReLU(0.2), no norm,
AvgPool2d(3, 2, 1) down, bilinear x2 align_corners=True up, skip concatenation [encoder feature, up-sampled], 1x1 output
convolution without activation).  Weights are keyed like the reference state_dict.  Pinned by
tests/golden/fullstep_hourglass_b2_32x48_usecnn_gap2.npz (the real reference's step with --use_cnn)."""
import torch.nn.functional as F


def _block(sd, prefix, x):
    for i in (0, 1):
        x = F.pad(x, (1, 1, 1, 1), mode='reflect')
        x = F.leaky_relu(F.conv2d(x, sd['%s.model.%d.conv.weight' % (prefix, i)], sd['%s.model.%d.conv.bias' % (prefix, i)]), 0.2)
    return x


def unet_forward(sd, x, n_down=3):
    feat = []
    for k in range(n_down):
        x = _block(sd, 'down_%02d' % k, x)
        feat.append(x)
        x = F.avg_pool2d(x, 3, 2, 1)
    x = _block(sd, 'mid_conv', x)
    for k in range(n_down):
        up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
        x = _block(sd, 'up_%04d' % k, torch_cat(feat[-(k + 1)], up))
    return F.conv2d(x, sd['output_conv.conv.weight'], sd['output_conv.conv.bias'])


def torch_cat(a, b):
    import torch
    return torch.cat([a, b], 1)


def sf_net(opt, sd, P, ts):
    """Model.forward_sf_net with --use_cnn (models/scene_flow_motion_field.py:346-357)."""
    x = torch_cat(P, ts) if opt.time_dependent else P
    return unet_forward(sd, x, int(getattr(opt, 'n_down', 3))) / opt.sf_mag_div
