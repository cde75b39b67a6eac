"""Oracle (test-only): masked reprojection losses and the acceleration
regulariser, as a functional restatement of

  * Model.disp_loss   /root/reference/models/scene_flow_motion_field.py:140-150
  * Model._calc_loss  :285-324
  * Model._opt_reg    :326-344
  * Model._predict_on_batch(is_train=True)  :229-264  (geometry + MLP part;
    the depth maps are inputs here so the warp+loss path can be checked with
    leaf depths, SURVEY.md section 8d "stand-alone warp+loss microbench")
"""

from types import SimpleNamespace

import torch

from . import geometry as G
from . import sceneflow_mlp as M


def default_opt(**over):
    """The shipped flag set (experiments/davis/train_sequence.sh:24-63)."""
    o = dict(midas=True, use_disp=True, use_disp_ratio=False, time_dependent=True,
             flow_mul=1.0, disp_mul=1.0, acc_mul=1.0, sf_mag_div=100.0, interp_steps=5,
             warm_reg=False, weight_steps=False, use_motion_seg=False,
             n_freq_xyz=16, n_freq_t=16, use_cnn=False, n_down=3)
    o.update(over)
    return SimpleNamespace(**o)


def disparity_error(opt, d1, d2):
    if opt.use_disp:
        a = torch.clamp(d1, min=1e-3)
        b = torch.clamp(d2, min=1e-3)
        return 100 * torch.abs((1 / a) - (1 / b))
    if opt.use_disp_ratio:
        a = torch.clamp(d1, min=1e-3)
        b = torch.clamp(d2, min=1e-3)
        return torch.max(a, b) / torch.min(a, b) - 1
    return torch.abs(d1 - d2)


def valid_mask(opt, mask_2, depth_1, warped_p2_camera_2):
    """mask_2 * [depth_1 < 100] * [W2.z < 100]   (:286-289; midas only)."""
    m = mask_2
    if opt.midas:
        m = (depth_1 < 100).float().squeeze(1)[..., None, None] * m
        m = (warped_p2_camera_2[..., 2] < 100).float().squeeze(3)[..., None, None] * m
    return m


def masked_losses(opt, warm, mask_2, flow_1_2, depth_1, dflow, p1_camera_2,
                  warped_p2_camera_2, sf_by_depth, sf_1_2):
    """Returns (loss, parts dict, occ_mask).  Same reductions as :291-319."""
    m = valid_mask(opt, mask_2, depth_1, warped_p2_camera_2)
    if warm:
        per_px = torch.nn.functional.mse_loss(dflow, flow_1_2, reduction='none')
    else:
        per_px = torch.nn.functional.l1_loss(dflow, flow_1_2, reduction='none')
    occ = m[:, None, ..., 0, 0].permute([0, 2, 3, 1])               # [B,H,W,1]
    denom = torch.sum(occ) + 1e-8
    flow_loss = torch.sum(occ * per_px.squeeze(3)) / denom
    disp_pp = disparity_error(opt, p1_camera_2[..., -1], warped_p2_camera_2[..., -1]).permute([0, 3, 1, 2])
    disp_loss = torch.sum(occ[:, None, ..., 0] * disp_pp[:, 0:1, ...]) / denom
    sf_pp = torch.abs(sf_by_depth.squeeze(3).permute(0, 3, 1, 2) - sf_1_2)
    sf_loss = torch.sum(occ[:, None, ..., 0] * sf_pp) / denom
    second = disp_loss if opt.use_disp else sf_loss
    loss = flow_loss * opt.flow_mul + second * opt.disp_mul
    return loss, {'flow_loss_1_2': flow_loss, 'disp_loss_1_2': disp_loss, 'sf_loss': sf_loss,
                  'mask_sum': torch.sum(occ), 'sf_loss_pp': sf_pp.sum(1).detach()}, occ


def integer_steps(time_stamp_1, time_stamp_2, time_step):
    gap = torch.mean(time_stamp_2 - time_stamp_1)
    return int((gap / time_step).round().long().item())


def predict_train(opt, sd_mlp, batch, depth_1, depth_2):
    """Geometry + MLP half of _predict_on_batch(is_train=True)."""
    cams = {k: batch[k] for k in ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')}
    st = G.static_reprojection(depth_1, depth_2, batch['flow_1_2'], **cams)
    P1 = st['global_p1'].squeeze(3).permute(0, 3, 1, 2)                 # B3HW view
    time_step = float(batch['time_step'].squeeze().item())
    steps = integer_steps(batch['time_stamp_1'], batch['time_stamp_2'], time_step)
    kw = dict(n_freq_xyz=opt.n_freq_xyz, n_freq_t=opt.n_freq_t)
    ts = batch['time_stamp_1'] if opt.time_dependent else None
    if getattr(opt, 'use_cnn', False):                 # the U-Net scene-flow network (oracle/fcn_unet.py), same Euler loop
        from . import fcn_unet as U
        sf, p, tcur = 0, P1, batch['time_stamp_1']
        for _ in range(steps):
            s = U.sf_net(opt, sd_mlp, p, tcur)
            sf, p, tcur = sf + s, p + s, tcur + time_step
    elif opt.time_dependent:
        sf = M.sf_multi_step(sd_mlp, P1, ts, time_step, steps, opt.sf_mag_div, **kw)
    else:
        sf = 0
        p = P1
        for _ in range(steps):
            s = M.mlp_forward(sd_mlp, p, None, **kw) / opt.sf_mag_div
            sf = sf + s
            p = p + s
    if opt.use_motion_seg:
        sf = sf * batch['motion_seg_1'].squeeze(3).permute(0, 3, 1, 2)
    sflow = sf.permute(0, 2, 3, 1)[..., None, :]
    dyn = G.dynamic_reprojection(depth_1, depth_2, batch['flow_1_2'], batch['flow_2_1'],
                                 sflow_1_2=sflow, sflow_2_1=sflow, **cams)
    dyn['sf_1_2'] = sf
    dyn['global_p1'] = P1
    dyn['sf_by_dep_1_2'] = st['sf_by_depth']
    dyn['_steps'] = steps
    dyn['_static'] = st
    return dyn


def train_losses(opt, warm, batch, pred):
    return masked_losses(opt, warm, batch['mask_2'], batch['flow_1_2'], pred['depth_1'],
                         pred['dflow_1_2'], pred['p1_camera_2'], pred['warped_p2_camera_2'],
                         pred['sf_by_dep_1_2'], pred['sf_1_2'])


def acceleration_reg(opt, sd_mlp, batch, P1):
    """_opt_reg: acc_mul * mean |sf(P1+sf0, t+dt) - sf0|  (un-detached P1)."""
    kw = dict(n_freq_xyz=opt.n_freq_xyz, n_freq_t=opt.n_freq_t)
    time_step = float(batch['time_step'].squeeze().item())
    ts = batch['time_stamp_1'] if opt.time_dependent else None
    if getattr(opt, 'use_cnn', False):
        from . import fcn_unet as U
        sf0 = U.sf_net(opt, sd_mlp, P1, batch['time_stamp_1'])
        sf1 = U.sf_net(opt, sd_mlp, P1 + sf0, batch['time_stamp_1'] + time_step)
        ones = torch.ones_like(sf0)
        return (ones * torch.abs(sf1 - sf0)).sum() / (ones.sum() + 1e-6) * opt.acc_mul
    sf0 = M.mlp_forward(sd_mlp, P1, ts, **kw) / opt.sf_mag_div
    ones = torch.ones_like(sf0)
    ts1 = (ts + time_step) if ts is not None else None
    sf1 = M.mlp_forward(sd_mlp, P1 + sf0, ts1, **kw) / opt.sf_mag_div
    acc = (ones * torch.abs(sf1 - sf0)).sum() / (ones.sum() + 1e-6)
    return acc * opt.acc_mul


def warp_loss_with_leaf_depths(opt, warm, sd_mlp, batch, depth_1, depth_2, with_reg=None):
    """Everything downstream of the depth nets for one step, on leaf depths.

    Returns dict with the loss scalars, pred surfaces and gradients w.r.t.
    depth_1, depth_2 and every MLP parameter (main loss and, when enabled,
    the regulariser accumulated on top, exactly like the two .backward()
    calls at scene_flow_motion_field.py:193-195).
    """
    d1 = depth_1.detach().clone().requires_grad_(True)
    d2 = depth_2.detach().clone().requires_grad_(True)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in sd_mlp.items()}
    pred = predict_train(opt, sd, batch, d1, d2)
    loss, parts, occ = train_losses(opt, warm, batch, pred)
    if opt.weight_steps:
        loss = loss * pred['_steps']
    do_reg = (opt.interp_steps > 0 and (not warm or opt.warm_reg) and opt.acc_mul > 0) \
        if with_reg is None else with_reg
    out = {'loss': loss.detach(), 'parts': {k: v.detach() for k, v in parts.items()}, 'pred': pred,
           'occ': occ.detach()}
    if do_reg:
        loss.backward(retain_graph=True)
        reg = acceleration_reg(opt, sd, batch, pred['global_p1'])
        reg.backward()
        out['acc_reg'] = reg.detach()
    else:
        loss.backward()
        out['acc_reg'] = torch.zeros(())
    out['g_depth_1'] = d1.grad if d1.grad is not None else torch.zeros_like(d1)
    out['g_depth_2'] = d2.grad if d2.grad is not None else torch.zeros_like(d2)
    out['g_mlp'] = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    return out


def warp_loss_leaf_sf(opt, warm, batch, depth_1, depth_2, sf_1_2, need_grads=True):
    """The warp+loss operator in isolation: scene flow is a leaf [B,3,H,W]
    (no MLP), exactly the contract of dvd_warp_loss_fused.  Returns the four
    un-normalised sums, the normalised losses and autograd gradients of the
    *normalised* loss w.r.t. depth_1, depth_2, sf_1_2."""
    d1 = depth_1.detach().clone().requires_grad_(need_grads)
    d2 = depth_2.detach().clone().requires_grad_(need_grads)
    sf = sf_1_2.detach().clone().requires_grad_(need_grads)
    cams = {k: batch[k] for k in ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')}
    st = G.static_reprojection(d1, d2, batch['flow_1_2'], **cams)
    sflow = sf.permute(0, 2, 3, 1)[..., None, :]
    dyn = G.dynamic_reprojection(d1, d2, batch['flow_1_2'], batch['flow_2_1'], sflow_1_2=sflow,
                                 sflow_2_1=sflow, **cams)
    loss, parts, occ = masked_losses(opt, warm, batch['mask_2'], batch['flow_1_2'], dyn['depth_1'],
                                     dyn['dflow_1_2'], dyn['p1_camera_2'], dyn['warped_p2_camera_2'],
                                     st['sf_by_depth'], sf)
    out = {'loss': loss.detach(), 'parts': {k: v.detach() for k, v in parts.items()}, 'occ': occ.detach(),
           'behind': dyn['_behind'].detach(), 'dyn': dyn, 'static': st}
    S0 = parts['mask_sum'].detach()
    den = S0 + 1e-8
    out['sums'] = torch.stack([S0, parts['flow_loss_1_2'].detach() * den, parts['disp_loss_1_2'].detach() * den,
                               parts['sf_loss'].detach() * den])
    if need_grads:
        loss.backward()
        z = torch.zeros_like
        out['g_depth_1'] = d1.grad if d1.grad is not None else z(d1)
        out['g_depth_2'] = d2.grad if d2.grad is not None else z(d2)
        out['g_sf'] = sf.grad if sf.grad is not None else z(sf)
    return out
