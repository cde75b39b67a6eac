"""The CPU oracle must reproduce the fixtures generated from the real
reference (tests/golden/make_golden.py) -- this is what pins the oracle.

Same host + same torch build => the restatement runs the same ATen kernels in
the same order and is bit-identical; across hosts (the GPU box may dispatch a
different CPU capability) we allow a few ulp on floats and still demand
bit-exact index masks.
"""
import numpy as np
import pytest
import torch

from helpers import CAM_KEYS, golden_batch, golden_mlp_sd, golden_opt, load_golden, t
from oracle import geometry as G
from oracle import losses as L
from oracle import sceneflow_mlp as M

TIGHT = dict(rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize('name', ['geom_b2_24x32', 'geom_b3_16x40_behind'])
def test_geometry_surfaces(name):
    gd = load_golden(name)
    cams = {k: t(gd['in_' + k]) for k in CAM_KEYS}
    d1, d2, flow = t(gd['in_depth_1']), t(gd['in_depth_2']), t(gd['in_flow_1_2'])
    sf = t(gd['in_sf_1_2'])
    st = G.static_reprojection(d1, d2, flow, **cams)
    for k in ('dflow_1_2', 'sf_by_depth', 'warped_global_p2', 'global_p1'):
        np.testing.assert_allclose(st[k].numpy(), gd['fbd_' + k], **TIGHT, err_msg=k)
    sflow = sf.permute(0, 2, 3, 1)[..., None, :]
    dy = G.dynamic_reprojection(d1, d2, flow, -flow, sflow_1_2=sflow, sflow_2_1=sflow, **cams)
    for k in ('dflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'global_p1', 'staticflow_1_2', 'p1_camera_2',
              'warped_p2_camera_2'):
        np.testing.assert_allclose(dy[k].numpy(), gd['slack_' + k], **TIGHT, err_msg=k)
    # index masks, bit exact: behind-camera pixels are exactly those with zero predicted flow
    behind = dy['_behind'][..., 0, 0].numpy()
    gold_behind = (gd['slack_depth_image_1_2'][:, 0] < 1e-3)
    assert np.array_equal(behind, gold_behind)
    if 'behind' in name:
        assert behind.any() and not behind.all()
    np.testing.assert_allclose(G.unproject(d1, cams['R_1'], cams['t_1'], cams['K_inv']).numpy(),
                               gd['unproject_global_p1'], **TIGHT)
    grid = G.pixel_grid(d1.shape[2], d1.shape[3])
    np.testing.assert_allclose(G.flow_sample(d2, flow, grid).numpy(), gd['bwarp_depth_2'], **TIGHT)


def test_mlp_forward_backward():
    gd = load_golden('mlp_b2_8x16')
    sd = {k: v.requires_grad_(True) for k, v in golden_mlp_sd(gd).items()}
    x = t(gd['in_x']).requires_grad_(True)
    y = M.mlp_forward(sd, x, t(gd['in_t']))
    np.testing.assert_allclose(y.detach().numpy(), gd['out_y'], rtol=1e-5, atol=1e-6)
    (y * t(gd['up_y'])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), gd['g_x'], rtol=1e-4, atol=1e-6)
    for k, v in sd.items():
        g = gd['gsd_' + k]
        np.testing.assert_allclose(v.grad.numpy(), g, rtol=1e-4, atol=1e-5 * np.abs(g).max(), err_msg=k)


def test_mlp_init_statistics():
    sd = M.init_params(seed=0)
    dims = M.layer_dims()
    assert dims == [132, 256, 256, 256, 256, 256, 3]
    assert sum(v.numel() for v in sd.values()) == 297987
    w = sd['convs.1.conv.weight']
    assert abs(float(w.std()) - (2.0 / (1 + 0.04)) ** 0.5 / 16.0) < 2e-3


@pytest.mark.parametrize('name', ['step_b2_24x32_full', 'step_b2_24x32_warm', 'step_b3_16x40_behind_gap2',
                                  'step_b2_16x24_sfloss', 'step_b2_16x24_ratio'])
def test_step_losses_and_grads(name):
    gd = load_golden(name)
    opt = golden_opt(gd)
    warm = bool(gd['warm'])
    batch = golden_batch(gd)
    sd = golden_mlp_sd(gd)
    out = L.warp_loss_with_leaf_depths(opt, warm, sd, batch, t(gd['in_depth_1']), t(gd['in_depth_2']))
    assert out['pred']['_steps'] == int(gd['steps'])
    for k in ('flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'):
        np.testing.assert_allclose(float(out['parts'][k]), float(gd['loss_' + k]), rtol=1e-6, err_msg=k)
    np.testing.assert_allclose(float(out['loss']), float(gd['loss_loss']), rtol=1e-6)
    np.testing.assert_allclose(float(out['acc_reg']), float(gd['loss_acc_reg']), rtol=1e-5, atol=1e-9)
    for k in ('dflow_1_2', 'p1_camera_2', 'warped_p2_camera_2', 'sf_1_2', 'global_p1', 'sf_by_dep_1_2'):
        np.testing.assert_allclose(out['pred'][k].detach().numpy(), gd['pred_' + k], rtol=1e-5, atol=1e-5, err_msg=k)
    for k in ('g_depth_1', 'g_depth_2'):
        g = gd[k]
        np.testing.assert_allclose(out[k].numpy(), g, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(g).max()), err_msg=k)
    for k, v in out['g_mlp'].items():
        g = gd['gsd_' + k]
        np.testing.assert_allclose(v.numpy(), g, rtol=1e-3, atol=1e-5 * np.abs(g).max(), err_msg=k)
    # valid-pixel mask is bit exact
    m_gold = (gd['in_mask_2'][..., 0, 0] * (gd['in_depth_1'][:, 0] < 100) *
              (gd['pred_warped_p2_camera_2'][..., 0, 2] < 100)).astype(np.float32)
    assert np.array_equal(out['occ'][..., 0].numpy(), m_gold)


@pytest.mark.parametrize('name', ['fullstep_hourglass_b2_32x48_train', 'fullstep_hourglass_b2_32x48_warm',
                                  'fullstep_hourglass_b2_32x48_mseg_gap2', 'fullstep_hourglass_b2_32x48_usecnn_gap2'])
def test_full_step_oracle_reproduces_the_reference_logs(name):
    """oracle.train_step (what bench.py times as cpu_baseline) against the batch_log the REAL reference
    Model._train_on_batch produced for the same seeded weights and batch (tests/golden/make_golden.py)."""
    import helpers
    from dvd_hip import synthetic
    from dvd_hip.networks.sceneflow_field import SceneFlowFieldNet
    from dvd_hip.third_party.hourglass import HourglassModel_Embed
    from oracle import train_step as T
    gd = load_golden(name)
    o = dict(helpers.FULL_STEP_OPT)
    if 'over_keys' in gd:
        o.update({str(k): (bool(v) if isinstance(o.get(str(k)), bool) else float(v))
                  for k, v in zip(gd['over_keys'], gd['over_vals'])})
    opt = L.default_opt(**{k: o[k] for k in ('midas', 'use_disp', 'use_disp_ratio', 'time_dependent', 'flow_mul', 'disp_mul',
                                             'acc_mul', 'sf_mag_div', 'interp_steps', 'warm_reg', 'weight_steps',
                                             'use_motion_seg', 'n_freq_xyz', 'n_freq_t', 'use_cnn', 'n_down')})
    seed = int(gd['seed'])
    depth = helpers.seeded_fill_(HourglassModel_Embed(noexp=False, use_embedding=False), seed)
    if opt.use_cnn:          # the U-Net scene-flow network: oracle/fcn_unet.py, weights keyed like the reference state_dict
        from dvd_hip.networks.FCNUnet import FCNUnet
        mlp = helpers.seeded_fill_(FCNUnet(None, n_down=3, feat=32, block_type='double_conv', in_channel=4, out_channel=3), seed + 1)
    else:
        mlp = helpers.seeded_fill_(SceneFlowFieldNet(net_width=256, n_layers=4, time_dependent=True, N_freq_xyz=16,
                                                     N_freq_t=16), seed + 1)
    sd = T.mlp_state_from_module(mlp)
    batch = synthetic.make_batch(int(gd['B']), int(gd['H']), int(gd['W']), gap=int(gd['gap']), seed=seed + 2)
    warm = int(gd['epoch']) <= o['warm_sf']
    log, _ = T.train_step(opt, depth, sd, batch, warm, lr_depth=o['lr'], lr_mlp=o['lr'] * o['scene_lr_mul'])
    for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'):
        np.testing.assert_allclose(log[k], float(gd['log_' + k]), rtol=1e-5, err_msg=k)
    np.testing.assert_allclose(log['acc_reg'], float(gd['log_acc_reg']), rtol=1e-4, atol=1e-10)


def test_flow_consistency_masks_match_the_reference_code():
    """oracle/preprocess.py against tests/golden/flow_masks.npz (the reference's own mask code, executed)."""
    import helpers
    from oracle import preprocess as OP
    gd = load_golden('flow_masks')
    for H, W, seed, noise in helpers.FLOW_MASK_CASES:
        f12, f21 = helpers.flow_pair(H, W, seed, noise)
        np.testing.assert_allclose(gd['flow_crc_%dx%d' % (H, W)], [float(f12.double().sum()), float(f21.double().sum())],
                                   rtol=1e-12)
        m1, m2 = OP.consistency_masks(f12.numpy(), f21.numpy())
        assert np.array_equal(np.packbits(m1), gd['mask_1_%dx%d' % (H, W)])
        assert np.array_equal(np.packbits(m2), gd['mask_2_%dx%d' % (H, W)])
