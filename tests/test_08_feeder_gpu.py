"""Pair-pack reader + host->HBM feeder: a synthetic video written in the `.pt` layout of
scripts/preprocess/davis/generate_sequence_midas.py:117-179 is read back by `Dataset` with the sample schema of
datasets/davis_sequence.py:98-115, fed through `DeviceFeeder`, and drives `train_epoch`; the step on a fed
batch equals the step on the same tensors placed on the device by hand."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _write_video(root, n_packs, B, H, W):
    from dvd_hip import synthetic
    from dvd_hip.datasets import davis_sequence as D
    seq = os.path.join(root, D.SEQ_PREFIX, 'dog', '001')
    frames = os.path.join(root, D.FRAME_PREFIX, 'dog')
    os.makedirs(seq)
    os.makedirs(frames)
    for f in range(100):
        np.savez(os.path.join(frames, 'frame_%05d.npz' % f), img=np.zeros((2, 2, 3), np.float32))
    batches = []
    for i in range(n_packs):
        b = synthetic.make_batch(B, H, W, gap=1 + i % 2, seed=50 + i)
        D.write_pair_pack(os.path.join(seq, 'shuffle_False_gap_%02d_sequence_%05d.pt' % (1 + i % 2, i)), b)
        batches.append(b)
    return batches


def test_reader_schema_and_feeder_drive_the_step(tmp_path):
    from dvd_hip.datasets import get_dataset
    from dvd_hip.datasets.davis_sequence import DeviceFeeder
    from dvd_hip.models.scene_flow_motion_field import Model
    B, H, W = 2, 32, 48
    batches = _write_video(str(tmp_path), 3, B, H, W)
    o = dict(helpers.FULL_STEP_OPT)
    o.update(full_logdir='/tmp', track_id='dog', gaps='1,2', repeat=1, subsample=False, overfit=False, data_root=str(tmp_path))
    opt = SimpleNamespace(**o)
    with pytest.warns(UserWarning):
        model = Model(opt, None)
    helpers.seeded_fill_(model.net_depth, 3)
    helpers.seeded_fill_(model.net_sceneflow, 4)
    model.to(torch.device('cuda'))
    ds = get_dataset('davis_sequence')(opt, mode='train', model=model)
    assert len(ds) == 3 and ds.n_frames == 100.0
    s = ds[0]
    # schema of datasets/davis_sequence.py:98-115
    assert s['img_1'].shape == (B, 3, H, W) and s['flow_1_2'].shape == (B, H, W, 2) and s['mask_2'].shape == (B, H, W, 1, 1)
    assert s['time_stamp_1'].shape == (B, 1, H, W) and abs(s['time_step'] - 0.01) < 1e-12 and s['frame_id_2'].shape == (B,)
    assert torch.equal(s['time_stamp_2'][:, 0, 0, 0], (batches[0]['frame_id_2'] / 100.0))
    assert set(model.input_names) & set(s) >= {'img_1', 'img_2', 'flow_1_2', 'mask_2', 'R_1', 'R_2_T', 't_2', 'K', 'K_inv',
                                              'time_stamp_1', 'time_stamp_2', 'time_step', 'motion_seg_1'}
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    fed = list(DeviceFeeder(loader, 'cuda'))
    assert len(fed) == 3 and fed[1]['img_1'].is_cuda and fed[1]['img_1'].shape == (1, B, 3, H, W)
    # gaps are listed per gap: files of gap 1 first (packs 0, 2), then gap 2 (pack 1)
    order = [0, 2, 1]
    for got, i in zip(fed, order):
        assert torch.equal(got['flow_1_2'][0].cpu(), batches[i]['flow_1_2'])
    # the step on a fed batch == the step on hand-placed tensors (fresh models, same weights)
    logs = []
    for src in ('fed', 'hand'):
        with pytest.warns(UserWarning):
            m = Model(opt, None)
        helpers.seeded_fill_(m.net_depth, 3)
        helpers.seeded_fill_(m.net_sceneflow, 4)
        m.to(torch.device('cuda'))
        if src == 'fed':
            batch = next(iter(DeviceFeeder(loader, 'cuda')))
        else:
            batch = helpers.loader_batch({k: (v.cuda() if k != 'time_step' else v) for k, v in batches[0].items()})
        logs.append(m._train_on_batch(6, 0, batch))
    for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg'):
        np.testing.assert_allclose(logs[0][k], logs[1][k], rtol=1e-6, err_msg=k)
    # and train_epoch runs off the feeder (mixed frame gaps across the packs)
    elog = model.train_epoch(DeviceFeeder(loader, 'cuda'), epochs=1, initial_epoch=6)
    assert np.isfinite(elog['loss']) and model._flat_sf.step_count == 3
