"""The rest of the Model surface the reference's train.py / test.py drive:
inference path (`test_on_batch`, `_vali_on_batch`: depth net + unprojection + one scene-flow evaluation,
models/scene_flow_motion_field.py:265-275, models/video_base.py:72-103,128-155), `train_epoch` with the
logger callbacks (models/netinterface.py:246-360) and the checkpoint round trip
(netinterface.py:528-562) including the fused-Adam state."""
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import helpers
from oracle import geometry as G
from oracle import sceneflow_mlp as M

pytestmark = pytest.mark.gpu


def _model(seed=41, **over):
    from dvd_hip.models.scene_flow_motion_field import Model
    o = dict(helpers.FULL_STEP_OPT)
    o.update(midas=False, full_logdir='/tmp', lr=1e-4)
    o.update(over)
    with pytest.warns(UserWarning):
        model = Model(SimpleNamespace(**o), None)
    helpers.seeded_fill_(model.net_depth, seed)
    helpers.seeded_fill_(model.net_sceneflow, seed + 1)
    cpu_depth = copy.deepcopy(model.net_depth).eval()
    cpu_sd = {k: v.detach().clone() for k, v in model.net_sceneflow.state_dict().items()}
    model.to(torch.device('cuda'))
    return model, cpu_depth, cpu_sd


def _inference_batch(B, H, W, seed=3):
    from dvd_hip import synthetic
    b = synthetic.make_batch(B, H, W, gap=1, seed=seed)
    out = {'img': b['img_1'], 'R_1': b['R_1'], 't_1': b['t_1'], 'K_inv': b['K_inv'], 'time_stamp_1': b['time_stamp_1'],
           'frame_id_1': b['frame_id_1']}
    g = torch.Generator().manual_seed(seed)
    out['depth_mvs'] = 1.0 + 4.0 * torch.rand(B, 1, H, W, generator=g)
    out['depth_mvs'][:, :, :2] = 0.0                      # invalid MVS depth: excluded by the validity mask
    return out


def test_inference_path_matches_cpu():
    model, cpu_depth, cpu_sd = _model()
    batch = _inference_batch(2, 32, 48)
    import os
    import tempfile
    model.opt.output_dir, model.opt.epoch = tempfile.mkdtemp(), 7
    batch['pair_path'] = ['a', 'b']
    pred = model.test_on_batch(0, batch)
    # video_base.py:105-155: packed output, cached and written as <output_dir>/epoch0007_test/batch0000.npz
    assert {'batch_size', 'img_1', 'img_2', 'depth', 'sf_1_2', 'depth_gt', 'pair_path'} <= set(pred)
    assert pred['batch_size'] == 2 and pred['depth'].shape == (2, 1, 32, 48) and pred['sf_1_2'].shape == (2, 3, 32, 48)
    saved = np.load(os.path.join(model.opt.output_dir, 'epoch0007_test', 'batch0000.npz'))
    np.testing.assert_array_equal(saved['depth'], pred['depth'])
    assert len(model.test_cache) == 1 and model.outdir.endswith('epoch0007_test')
    with torch.no_grad():
        d = cpu_depth(batch['img'], batch['frame_id_1'].long())
        P = G.unproject(d, batch['R_1'], batch['t_1'], batch['K_inv']).squeeze(3).permute(0, 3, 1, 2)
        sf = M.mlp_forward(cpu_sd, P, batch['time_stamp_1']) / model.opt.sf_mag_div
    np.testing.assert_allclose(pred['depth'], d.numpy(), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(pred['sf_1_2'], sf.numpy(), rtol=1e-3, atol=2e-6)
    # validation metric: disparity MSE on the valid MVS pixels (video_base.py:72-103)
    log = model._vali_on_batch(1, 0, batch)
    gt = batch['depth_mvs']
    valid = (gt > 1e-2).float()
    disp = lambda z: (1 / (z + (1 - (z > 1e-2).float()) * 1e-8)) * (z > 1e-2).float()
    want = torch.nn.functional.mse_loss(disp(d) * valid, disp(gt) * valid).item()
    assert log['size'] == 2
    np.testing.assert_allclose(log['loss'], want, rtol=1e-3)


def test_train_epoch_drives_the_step_and_the_logger_callbacks():
    from dvd_hip import synthetic
    model, _, _ = _model()
    loader = [helpers.loader_batch(synthetic.make_batch(2, 32, 48, gap=1, seed=s)) for s in (1, 2, 3)]   # B=2: the step strips the loader dim in place
    elog = model.train_epoch(loader, epochs=2, initial_epoch=6, max_batches_per_train=2)
    logger = model._logger
    assert len(logger.batch_logs) == 4 and [e for e, _ in logger.epoch_logs] == [6, 7]
    for log in logger.batch_logs:
        assert {'loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg', 'batch', 'epoch', 'data_time'} <= set(log)
        assert np.isfinite(log['loss'])
    np.testing.assert_allclose(elog['loss'], np.mean([l['loss'] for l in logger.batch_logs[2:]]), rtol=1e-6)
    assert model._flat_sf.step_count == 4 and model._flat_depth.step_count == 4
    assert model.num_parameters() == sum(p.numel() for n in model._nets for p in n.parameters())


def test_checkpoint_round_trip_restores_weights_and_adam_state(tmp_path):
    from dvd_hip import synthetic
    batch = synthetic.make_batch(2, 32, 48, gap=1, seed=9)
    a, _, _ = _model(seed=51)
    a._train_on_batch(6, 0, helpers.loader_batch(dict(batch)))
    path = str(tmp_path / 'ckpt.pt')
    a.save_state_dict(path, save_optimizer=True, additional_values={'epoch': 6})
    sd = torch.load(path, map_location='cpu')
    assert set(sd) == {'nets', 'optimizers', 'epoch'} and 'convs.0.conv.weight' in sd['nets'][1]
    b, _, _ = _model(seed=77)                               # different weights until the checkpoint is loaded
    extra = b.load_state_dict(path)
    assert extra == {'epoch': 6} and b._flat_sf.step_count == 1
    assert torch.equal(a._flat_depth.flat, b._flat_depth.flat) and torch.equal(a._flat_sf.exp_avg_sq, b._flat_sf.exp_avg_sq)
    la = a._train_on_batch(6, 1, helpers.loader_batch(dict(batch)))
    lb = b._train_on_batch(6, 1, helpers.loader_batch(dict(batch)))
    for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg'):
        np.testing.assert_allclose(la[k], lb[k], rtol=1e-6, err_msg=k)
    assert float((a._flat_sf.flat - b._flat_sf.flat).abs().max()) <= 1e-6


REFERENCE_PRED_KEYS = {'dflow_1_2', 'depth_image_1_2', 'depth_warp_1_2', 'depth_1', 'depth_2', 'scenef_1_2', 'global_p1',
                       'staticflow_1_2', 'p1_camera_2', 'warped_p2_camera_2', 'sf_1_2', 'sf_by_dep_1_2', 'sf_loss_pp'}


def test_train_time_pred_dict_and_visual_export(tmp_path):
    """The reference materialises thirteen per-pixel surfaces every step and exports them with np.savez on the
    visualisation batches (scene_flow_motion_field.py:201-225, 243-264).  The fused step does not build them; on
    demand they must have the reference's keys and shapes and the oracle's values for this step's depths / weights."""
    from dvd_hip import synthetic
    from oracle import losses as L
    model, _, cpu_sd = _model(seed=61, use_motion_seg=True, vis_every_train=1, vis_at_start=True, vis_batches_train=0,
                              full_logdir=str(tmp_path))
    batch = synthetic.make_batch(2, 32, 48, gap=2, seed=8)
    model._train_on_batch(6, 0, helpers.loader_batch(dict(batch)))
    pred = model._predict_on_batch(is_train=True)
    assert set(pred) == REFERENCE_PRED_KEYS
    opt = L.default_opt(use_motion_seg=True, midas=False)
    want = L.predict_train(opt, cpu_sd, batch, model._last['depth_1'].cpu(), model._last['depth_2'].cpu())
    want['sf_loss_pp'] = torch.abs(want['sf_by_dep_1_2'].squeeze(3).permute(0, 3, 1, 2) - want['sf_1_2']).sum(1)
    for k in sorted(REFERENCE_PRED_KEYS):
        got, ref = pred[k].cpu(), want[k].detach()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-3, atol=2e-4, err_msg=k)
    out = np.load(str(tmp_path / 'visualize' / 'epoch0006_train' / 'rank0000_batch0000.npz'))
    assert REFERENCE_PRED_KEYS | {'batch_size', 'img_1', 'img_2', 'flow_1_2', 'flow_2_1'} <= set(out.files)
    np.testing.assert_array_equal(out['dflow_1_2'], pred['dflow_1_2'].cpu().numpy())
