#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (it imports /root/reference, which does not
exist on the GPU box):

    python tests/golden/make_golden.py

It drives the unmodified reference classes
  losses.scene_flow_projection.{flow_by_depth, scene_flow_projection_slack,
                                unproject_ptcld, BackwardWarp}
  networks.sceneflow_field.SceneFlowFieldNet
  models.scene_flow_motion_field.Model.{_predict_on_batch,_calc_loss,_opt_reg,
                                        forward_sf_net*,disp_loss}
on small seeded inputs and stores inputs + outputs (+ autograd gradients) as
.npz.  The Model methods are called on an instance created with
`Model.__new__` whose depth net is a stub returning fixed leaf depth maps, so
no checkpoint, torch.hub or logger is needed; every line of arithmetic that
runs is the reference's own.

Nothing here is product code and nothing in the repo imports this file.
"""

import os
import sys
from functools import partial
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
# where the fixtures are written: this directory, or $DVD_GOLDEN_OUT (tests/test_fixtures_regenerate_cpu.py writes a second
# set into a scratch directory and compares it with the committed one, array by array, bit for bit)
OUT_DIR = os.environ.get('DVD_GOLDEN_OUT') or HERE
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
sys.path.insert(0, ROOT)          # oracle/ (the independent ResNeXt restatement)

from dvd_hip import synthetic  # noqa: E402  (pure-torch input generator, no HIP needed)


def np_dict(d, prefix=''):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            out[prefix + k] = v.detach().cpu().numpy()
    return out


def ref_modules():
    from losses.scene_flow_projection import (flow_by_depth, scene_flow_projection_slack,
                                              unproject_ptcld, BackwardWarp)
    from networks.sceneflow_field import SceneFlowFieldNet
    return flow_by_depth, scene_flow_projection_slack, unproject_ptcld, BackwardWarp, SceneFlowFieldNet


CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')


def case_geometry(name, B, H, W, gap, behind, seed):
    fbd, slack, unproj, bwarp, _ = ref_modules()
    batch = synthetic.make_batch(B, H, W, gap=gap, seed=seed, behind_camera_pairs=behind, with_images=False)
    d1, d2 = synthetic.make_depths(B, H, W, seed=seed + 1, far_depth_frac=0.02)
    sf = synthetic.make_scene_flow(B, H, W, seed=seed + 2)
    cams = {k: batch[k] for k in CAM_KEYS}
    d1r = d1.clone().requires_grad_(True)
    d2r = d2.clone().requires_grad_(True)
    sfr = sf.clone().requires_grad_(True)
    st = fbd()(d1r, d2r, batch['flow_1_2'], **cams)
    sflow = sfr.permute(0, 2, 3, 1)[..., None, :]
    dy = slack()(d1r, d2r, batch['flow_1_2'], batch['flow_2_1'], sflow_1_2=sflow, sflow_2_1=sflow, **cams)
    out = {}
    out.update(np_dict({k: batch[k] for k in CAM_KEYS + ('flow_1_2', 'mask_2')}, 'in_'))
    out['in_depth_1'] = d1.numpy()
    out['in_depth_2'] = d2.numpy()
    out['in_sf_1_2'] = sf.numpy()
    out.update(np_dict(st, 'fbd_'))
    out.update(np_dict(dy, 'slack_'))
    # fixed upstream gradients -> gradients of the module outputs (operator-level backward check)
    g = torch.Generator().manual_seed(seed + 3)
    up = {}
    total = 0
    for key, t in list(st.items()) + [('s_' + k, v) for k, v in dy.items()]:
        if not t.requires_grad or key in ('s_depth_1', 's_depth_2', 's_scenef_1_2'):
            continue
        u = torch.randn(t.shape, generator=g)
        up[key] = u
        total = total + (t * u).sum()
    total.backward()
    out.update(np_dict(up, 'up_'))
    out['g_depth_1'] = d1r.grad.numpy()
    out['g_depth_2'] = d2r.grad.numpy()
    out['g_sf_1_2'] = sfr.grad.numpy()
    # stand-alone unproject + BackwardWarp
    out['unproject_global_p1'] = unproj()(d1, batch['R_1'], batch['t_1'], batch['K_inv']).numpy()
    out['bwarp_depth_2'] = bwarp()(d2, batch['flow_1_2']).numpy()
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print('wrote', name, {k: v.shape for k, v in out.items() if k.startswith('slack_')})


def case_mlp(name, B, H, W, seed):
    *_, Net = ref_modules()
    torch.manual_seed(seed)
    net = Net(net_width=256, n_layers=4, time_dependent=True, N_freq_xyz=16, N_freq_t=16)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight.data, a=0.2, mode='fan_in')
            torch.nn.init.normal_(m.bias.data, 0.0, 0.05)   # non-zero so bias paths are exercised
    g = torch.Generator().manual_seed(seed + 1)
    x = (torch.randn(B, 3, H, W, generator=g) * 2.0).requires_grad_(True)
    t = torch.rand(B, 1, 1, 1, generator=g).expand(B, 1, H, W).contiguous()
    y = net(x, t)
    up = torch.randn(y.shape, generator=g)
    (y * up).sum().backward()
    out = {'in_x': x.detach().numpy(), 'in_t': t.numpy(), 'out_y': y.detach().numpy(), 'up_y': up.numpy(),
           'g_x': x.grad.numpy()}
    for k, v in net.state_dict().items():
        out['sd_' + k] = v.numpy()
    for k, p in net.named_parameters():
        out['gsd_' + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print('wrote', name)


def _fake_model(opt, batch, d1, d2, net):
    """A reference Model instance without NetInterface/loggers/checkpoints."""
    import inspect
    from models.scene_flow_motion_field import Model
    from losses.scene_flow_projection import flow_by_depth, scene_flow_projection_slack
    m = Model.__new__(Model)
    m.opt = opt
    m._input = SimpleNamespace(**batch)
    lookup = {id(batch['img_1']): d1, id(batch['img_2']): d2}
    m.net_depth = lambda img, *a: lookup[id(img)]
    m.net_sceneflow = net
    m.L1_crit = partial(F.l1_loss, reduction='none')
    m.L2_crit = partial(F.mse_loss, reduction='none')
    m.warp = scene_flow_projection_slack()
    m.depth_flow = flow_by_depth()
    m.warp_args = [a for a in inspect.getfullargspec(m.warp.forward).args[1:] if hasattr(m._input, a)]
    m.flow_args = [a for a in inspect.getfullargspec(m.depth_flow.forward).args[1:] if hasattr(m._input, a)]
    return m


def case_step(name, B, H, W, gap, behind, seed, warm, **opt_over):
    """Real _predict_on_batch + _calc_loss (+ _opt_reg) on leaf depths."""
    *_, Net = ref_modules()
    o = dict(midas=True, use_disp=True, use_disp_ratio=False, time_dependent=True, use_cnn=False,
             flow_mul=1.0, disp_mul=1.0, acc_mul=1.0, sf_mag_div=100.0, interp_steps=5,
             warm_reg=False, weight_steps=False, use_motion_seg=False, n_freq_xyz=16, n_freq_t=16)
    o.update(opt_over)
    opt = SimpleNamespace(**o)
    torch.manual_seed(seed)
    net = Net(net_width=256, n_layers=4, time_dependent=True, N_freq_xyz=16, N_freq_t=16)
    for mod in net.modules():
        if isinstance(mod, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(mod.weight.data, a=0.2, mode='fan_in')
            torch.nn.init.normal_(mod.bias.data, 0.0, 0.02)
    batch = synthetic.make_batch(B, H, W, gap=gap, seed=seed, behind_camera_pairs=behind)
    d1, d2 = synthetic.make_depths(B, H, W, seed=seed + 1, far_depth_frac=0.02)
    d1 = d1.requires_grad_(True)
    d2 = d2.requires_grad_(True)
    model = _fake_model(opt, batch, d1, d2, net)
    model.warm = warm
    pred = model._predict_on_batch()
    loss, loss_data = model._calc_loss(pred)
    do_reg = opt.interp_steps > 0 and (not warm or opt.warm_reg) and opt.acc_mul > 0
    if do_reg:
        loss.backward(retain_graph=True)
        acc = model._opt_reg(pred, steps=opt.interp_steps)
    else:
        loss.backward()
        acc = 0.0
    out = np_dict({k: v for k, v in batch.items() if k not in ('img_1', 'img_2')}, 'in_')
    out['in_depth_1'] = d1.detach().numpy()
    out['in_depth_2'] = d2.detach().numpy()
    out['opt_keys'] = np.array(sorted(o.keys()))
    out['opt_vals'] = np.array([float(o[k]) for k in sorted(o.keys())])
    out['warm'] = np.array(int(warm))
    out['steps'] = np.array(model.steps)
    for k, v in net.state_dict().items():
        out['sd_' + k] = v.numpy()
    for k, p in net.named_parameters():
        out['gsd_' + k] = p.grad.numpy()
    out['g_depth_1'] = d1.grad.numpy()
    out['g_depth_2'] = (d2.grad if d2.grad is not None else torch.zeros_like(d2)).numpy()
    for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'):
        out['loss_' + k] = np.array(loss_data[k], dtype=np.float64)
    out['loss_acc_reg'] = np.array(acc, dtype=np.float64)
    for k in ('dflow_1_2', 'p1_camera_2', 'warped_p2_camera_2', 'sf_1_2', 'global_p1', 'sf_by_dep_1_2',
              'staticflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'sf_loss_pp'):
        out['pred_' + k] = pred[k].detach().numpy()
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print('wrote', name, {k: float(out['loss_' + k]) for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg')})


def case_full_step(name, midas, B, H, W, gap, epoch, seed, over=None):
    """The REAL reference Model._train_on_batch (models/scene_flow_motion_field.py:152-227)
    on CPU: depth net (hourglass, or MiDaS whose `torch.hub.load` call -- unreachable here --
    returns oracle/resnext.py's independent restatement of torchvision's ResNeXt-101 32x8d; the reference's own
    `_make_pretrained_resnext101_wsl` / `_make_resnet_backbone` then run unmodified) + scene-flow MLP + warp + losses + both backward passes + Adam."""
    import tempfile
    import unittest.mock as mock
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    import third_party.hourglass as RH
    import third_party.midas_blocks as RB
    import third_party.MiDaS as RM
    import visualize.html_visualizer as HV
    from models.scene_flow_motion_field import Model
    from oracle import resnext          # independent restatement of torchvision's ResNeXt-101 32x8d (NOT the product's)
    o = dict(helpers.FULL_STEP_OPT)
    o.update(midas=midas, full_logdir=tempfile.mkdtemp())
    o.update(over or {})
    opt = SimpleNamespace(**o)

    class _Loggers(object):
        def add_logger(self, *a):
            pass

        def get_html_logger(self):
            return None
    real_load = torch.load
    with mock.patch.object(HV, 'Pool', lambda n: None), \
            mock.patch.object(torch.hub, 'load', lambda repo, entry, *a, **k: resnext.resnext101_32x8d()), \
            mock.patch.object(RM.BaseModel, 'load', lambda self, path: None), \
            mock.patch.object(torch, 'load', lambda path, *a, **k: RH.HourglassModel().state_dict()
                              if 'pretrained_depth_ckpt' in str(path) else real_load(path, *a, **k)):
        model = Model(opt, _Loggers())
    helpers.seeded_fill_(model.net_depth, seed)
    helpers.seeded_fill_(model.net_sceneflow, seed + 1)
    if midas:
        with torch.no_grad():
            model.net_depth.scratch.output_conv[4].weight.mul_(30.0)
            model.net_depth.scratch.output_conv[4].bias.fill_(2000.0)
    model.to(torch.device('cpu'))
    batch = synthetic.make_batch(B, H, W, gap=gap, seed=seed + 2)
    log = model._train_on_batch(epoch, 0, helpers.loader_batch(batch))
    out = {'B': np.array(B), 'H': np.array(H), 'W': np.array(W), 'gap': np.array(gap), 'epoch': np.array(epoch),
           'seed': np.array(seed), 'midas': np.array(int(midas)),
           'over_keys': np.array(sorted(over or {})), 'over_vals': np.array([float((over or {})[k]) for k in sorted(over or {})])}
    for k, v in log.items():
        out['log_' + k] = np.array(float(v), dtype=np.float64)
    names, gnorm, pnorm = [], [], []
    for prefix, net in (('depth', model.net_depth), ('sf', model.net_sceneflow)):
        for k, p in net.named_parameters():
            names.append(prefix + '/' + k)
            gnorm.append(0.0 if p.grad is None else float(p.grad.double().norm()))
            pnorm.append(float(p.data.double().norm()))
    out['param_names'] = np.array(names)
    out['grad_norms'] = np.array(gnorm)
    out['param_norms_after'] = np.array(pnorm)
    keep = ['convs.0.conv.weight', 'convs.3.conv.bias', 'convs.5.conv.weight', 'convs.5.conv.bias',
            'down_00.model.0.conv.weight', 'mid_conv.model.1.conv.weight', 'up_0002.model.0.conv.weight',
            'output_conv.conv.weight', 'output_conv.conv.bias']                 # (--use_cnn: the U-Net's keys)
    for k, p in model.net_sceneflow.named_parameters():
        if k in keep:
            out['g_sf/' + k] = p.grad.numpy()
            out['p_sf/' + k] = p.data.numpy()
    dkeep = (['scratch.output_conv.4.weight', 'scratch.output_conv.2.weight', 'pretrained.layer1.0.weight',
              'pretrained.layer4.2.bn3.weight'] if midas else
             ['net_depth.pred_layer.weight', 'net_depth.seq.0.weight', 'net_depth.seq.1.weight'])
    for k, p in model.net_depth.named_parameters():
        if k in dkeep and p.grad is not None:
            out['g_depth/' + k] = p.grad.numpy()
            out['p_depth/' + k] = p.data.numpy()
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print('wrote', name, {k: float(v) for k, v in log.items()})


def case_trajectory(name, midas, B, H, W, gap, epoch, seed, steps=5, over=None):
    """K consecutive `Model._train_on_batch` calls of the REAL reference (same construction as case_full_step) on ONE
    batch: the logged losses of every step, the norm of every parameter after step K and a few parameter tensors.  Pins
    more than one step: Adam's moments and bias correction across steps, the second step's forward on updated weights, and
    the direction the loss takes on a repeated batch."""
    import tempfile
    import unittest.mock as mock
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    import third_party.hourglass as RH
    import third_party.MiDaS as RM
    import visualize.html_visualizer as HV
    from models.scene_flow_motion_field import Model
    from oracle import resnext
    o = dict(helpers.FULL_STEP_OPT)
    o.update(midas=midas, full_logdir=tempfile.mkdtemp())
    o.update(over or {})
    opt = SimpleNamespace(**o)

    class _Loggers(object):
        def add_logger(self, *a):
            pass

        def get_html_logger(self):
            return None
    real_load = torch.load
    with mock.patch.object(HV, 'Pool', lambda n: None), \
            mock.patch.object(torch.hub, 'load', lambda repo, entry, *a, **k: resnext.resnext101_32x8d()), \
            mock.patch.object(RM.BaseModel, 'load', lambda self, path: None), \
            mock.patch.object(torch, 'load', lambda path, *a, **k: RH.HourglassModel().state_dict()
                              if 'pretrained_depth_ckpt' in str(path) else real_load(path, *a, **k)):
        model = Model(opt, _Loggers())
    helpers.seeded_fill_(model.net_depth, seed)
    helpers.seeded_fill_(model.net_sceneflow, seed + 1)
    if midas:
        with torch.no_grad():
            model.net_depth.scratch.output_conv[4].weight.mul_(30.0)
            model.net_depth.scratch.output_conv[4].bias.fill_(2000.0)
    model.to(torch.device('cpu'))
    batch = synthetic.make_batch(B, H, W, gap=gap, seed=seed + 2)
    out = {'B': np.array(B), 'H': np.array(H), 'W': np.array(W), 'gap': np.array(gap), 'epoch': np.array(epoch),
           'seed': np.array(seed), 'midas': np.array(int(midas)), 'steps': np.array(steps),
           'over_keys': np.array(sorted(over or {})), 'over_vals': np.array([float((over or {})[k]) for k in sorted(over or {})])}
    keys = ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg')
    series = {k: [] for k in keys}
    for i in range(steps):
        log = model._train_on_batch(epoch, i, helpers.loader_batch({k: (v.clone() if torch.is_tensor(v) else v)
                                                                    for k, v in batch.items()}))
        for k in keys:
            series[k].append(float(log[k]))
        print(name, 'step', i, {k: float(log[k]) for k in keys})
    for k in keys:
        out['series_' + k] = np.array(series[k], dtype=np.float64)
    names, pnorm = [], []
    for prefix, net in (('depth', model.net_depth), ('sf', model.net_sceneflow)):
        for k, p in net.named_parameters():
            names.append(prefix + '/' + k)
            pnorm.append(float(p.data.double().norm()))
    out['param_names'] = np.array(names)
    out['param_norms_after'] = np.array(pnorm)
    for k, p in model.net_sceneflow.named_parameters():
        if k in ('convs.0.conv.weight', 'convs.5.conv.weight', 'convs.5.conv.bias'):
            out['p_sf/' + k] = p.data.numpy()
    dkeep = (['scratch.output_conv.4.weight', 'scratch.output_conv.2.weight', 'pretrained.layer4.2.bn3.weight'] if midas else
             ['net_depth.pred_layer.weight', 'net_depth.seq.1.weight'])
    for k, p in model.net_depth.named_parameters():
        if k in dkeep:
            out['p_depth/' + k] = p.data.numpy()
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)
    print('wrote', name)


def case_flow_masks(name):
    """Occlusion / out-of-bounds masks: `get_oob_mask`, `backward_flow_warp` and the mask statements of
    `generate_pair_data` (scripts/preprocess/davis/generate_flows.py:57-82,139-148) cut out of the reference's source
    and executed as they are (tests/ref_exec.py) on the seeded flow pairs of tests/helpers.py.  Only the masks are
    stored; the inputs are regenerated from the seeds."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    import ref_exec
    out = {'cases': np.array(helpers.FLOW_MASK_CASES, dtype=np.float64)}
    for H, W, seed, noise in helpers.FLOW_MASK_CASES:
        f12, f21 = helpers.flow_pair(H, W, seed, noise)
        m1, m2 = ref_exec.reference_masks(f12.numpy(), f21.numpy())
        out['mask_1_%dx%d' % (H, W)] = np.packbits(m1)
        out['mask_2_%dx%d' % (H, W)] = np.packbits(m2)
        out['flow_crc_%dx%d' % (H, W)] = np.array([float(f12.double().sum()), float(f21.double().sum())])
        print(name, (H, W), 'masked fraction', m1.mean(), m2.mean())
    np.savez_compressed(os.path.join(OUT_DIR, name + '.npz'), **out)


# name -> how it is generated: ONE table, used by `python make_golden.py` (everything), `python make_golden.py <group>` and
# `python make_golden.py <fixture name> ...`
CASES = {
    'geom_b2_24x32': lambda n: case_geometry(n, B=2, H=24, W=32, gap=1, behind=0, seed=11),
    'geom_b3_16x40_behind': lambda n: case_geometry(n, B=3, H=16, W=40, gap=2, behind=1, seed=23),
    'mlp_b2_8x16': lambda n: case_mlp(n, B=2, H=8, W=16, seed=5),
    'step_b2_24x32_full': lambda n: case_step(n, B=2, H=24, W=32, gap=1, behind=0, seed=31, warm=False),
    'step_b2_24x32_warm': lambda n: case_step(n, B=2, H=24, W=32, gap=2, behind=0, seed=37, warm=True),
    'step_b3_16x40_behind_gap2': lambda n: case_step(n, B=3, H=16, W=40, gap=2, behind=1, seed=41, warm=False),
    'step_b2_16x24_sfloss': lambda n: case_step(n, B=2, H=16, W=24, gap=1, behind=0, seed=43, warm=False, use_disp=False),
    'step_b2_16x24_ratio': lambda n: case_step(n, B=2, H=16, W=24, gap=1, behind=0, seed=47, warm=False, use_disp=False,
                                               use_disp_ratio=True),
    'fullstep_hourglass_b2_32x48_train': lambda n: case_full_step(n, midas=False, B=2, H=32, W=48, gap=1, epoch=6, seed=101),
    'fullstep_hourglass_b2_32x48_warm': lambda n: case_full_step(n, midas=False, B=2, H=32, W=48, gap=2, epoch=1, seed=103),
    'fullstep_midas_b1_64x96_train': lambda n: case_full_step(n, midas=True, B=1, H=64, W=96, gap=1, epoch=6, seed=107),
    'fullstep_hourglass_b2_32x48_mseg_gap2': lambda n: case_full_step(n, midas=False, B=2, H=32, W=48, gap=2, epoch=6, seed=109,
                                                                      over=dict(use_motion_seg=True)),
    # BASELINE configs[0] shape (192x384, the reference's training resolution of record), MiDaS, 2 pairs
    'fullstep_midas_b2_192x384_train': lambda n: case_full_step(n, midas=True, B=2, H=192, W=384, gap=1, epoch=6, seed=113),
    # the U-Net scene-flow network (round 4)
    'fullstep_hourglass_b2_32x48_usecnn_gap2': lambda n: case_full_step(n, midas=False, B=2, H=32, W=48, gap=2, epoch=6, seed=127,
                                                                        over=dict(use_cnn=True)),
    'flow_masks': lambda n: case_flow_masks(n),
    # K-step trajectories of the real reference (round 5)
    'traj5_hourglass_b2_32x48': lambda n: case_trajectory(n, midas=False, B=2, H=32, W=48, gap=1, epoch=6, seed=131),
    # MiDaS with the learning rates of the shipped script (experiments/davis/train_sequence.sh:31,51: lr 1e-6, MLP x 1000).
    # With the one-step fixtures' lr = 1e-4 this 64 x 96 case is chaotic in the REFERENCE itself: two CPU runs of the real
    # reference that differ only in torch.set_num_threads (2 vs 8) agree to 1e-5 after the second step and to 2 % after the
    # third (loss 5.756 vs 5.641), 2.5 % after the fifth -- nothing to pin a port to; at the shipped rates the same two runs
    # agree to 1e-6 over all five steps.
    'traj5_midas_b1_64x96': lambda n: case_trajectory(n, midas=True, B=1, H=64, W=96, gap=1, epoch=6, seed=137,
                                                      over=dict(lr=1e-6, scene_lr_mul=1000.0)),
}
GROUPS = {
    'midas': ('fullstep_midas_b1_64x96_train', 'fullstep_midas_b2_192x384_train'),
    'use_cnn': ('fullstep_hourglass_b2_32x48_usecnn_gap2',),
    'trajectory': ('traj5_hourglass_b2_32x48', 'traj5_midas_b1_64x96'),
    'midas_192x384': ('fullstep_midas_b2_192x384_train',),
    # what tests/test_fixtures_regenerate_cpu.py regenerates on every CPU test run (about a minute): everything but the
    # large MiDaS steps
    'small': ('geom_b2_24x32', 'geom_b3_16x40_behind', 'mlp_b2_8x16', 'step_b2_24x32_full', 'step_b2_24x32_warm',
              'step_b3_16x40_behind_gap2', 'step_b2_16x24_sfloss', 'step_b2_16x24_ratio', 'fullstep_hourglass_b2_32x48_train',
              'fullstep_hourglass_b2_32x48_warm', 'fullstep_hourglass_b2_32x48_mseg_gap2', 'flow_masks',
              'traj5_hourglass_b2_32x48'),
}


def main():
    torch.set_num_threads(4)
    names = []
    for arg in sys.argv[1:]:
        names += list(GROUPS[arg]) if arg in GROUPS else [arg]
    for n in names or list(CASES):
        CASES[n](n)


if __name__ == '__main__':
    main()
