"""Interoperability with the REAL reference, run in the build container (skipped where /root/reference is absent):

  f2  the pair-pack reader: the reference's own writer (scripts/preprocess/davis/generate_sequence_midas.py
      `collate_sequence_fix_gap`, :78-170) writes a tiny video tree; the reference `datasets.davis_sequence.Dataset`
      (:22-154) and the product's read it; every key of every item is equal (train and vali modes), and the
      product's `write_pair_pack` emits the same pack layout as the reference writer.
  f3  the occlusion masks: the reference's own mask code (tests/ref_exec.py) against the oracle restatement and the
      committed fixture tests/golden/flow_masks.npz.
  f4  checkpoints: a file written by the reference `NetInterface.save_state_dict(save_optimizer=True)` after a real
      optimisation step (torch.optim.Adam state) loads into the product model, and the product's file loads back
      into the reference (`load_state_dict`, models/netinterface.py:528-562).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import helpers
import ref_exec

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason='reference checkout not present')


def _frame(H, W, i, rng, with_seg):
    th = 0.02 * i
    pose = np.eye(4)
    pose[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    pose[:3, 3] = [0.05 * i, 0.01 * i, 0.0]
    K = np.array([[0.9 * W, 0, (W - 1) / 2.0], [0, 0.9 * W, (H - 1) / 2.0], [0, 0, 1.0]])
    d = dict(img=rng.random((H, W, 3)).astype(np.float32), img_orig=rng.random((H, W, 3)).astype(np.float32),
             depth_pred=(1 + 4 * rng.random((H, W))).astype(np.float32), depth_mvs=(1 + 4 * rng.random((H, W))).astype(np.float32),
             pose_c2w=pose, intrinsics=K)
    if with_seg:
        d['motion_seg'] = (rng.random((H, W)) > 0.5).astype(np.float32)
    return d


def _write_tree(root, track, n_frames, H, W, with_seg):
    """frames_midas/<track>/frame_%05d.npz (layout of generate_frame_midas.py) and flow_pairs/<track>/flowpair_*.npz with
    masks computed by the reference's own mask statements (generate_flows.py:139-153)."""
    rng = np.random.default_rng(7)
    fdir = os.path.join(root, 'datafiles/davis_processed/frames_midas', track)
    pdir = os.path.join(root, 'datafiles/davis_processed/flow_pairs', track)
    os.makedirs(fdir)
    os.makedirs(pdir)
    for i in range(n_frames):
        np.savez(os.path.join(fdir, 'frame_%05d.npz' % i), **_frame(H, W, i, rng, with_seg))
    for gap in (1, 2):
        for i in range(n_frames - gap):
            f12, f21 = helpers.flow_pair(H, W, 100 * gap + i, 0.8)
            f12, f21 = f12.numpy(), f21.numpy()
            m1, m2 = ref_exec.reference_masks(f12, f21)
            np.savez(os.path.join(pdir, 'flowpair_%05d_%05d.npz' % (i, i + gap)), flow_1_2=f12, flow_2_1=f21, mask_1=m1,
                     mask_2=m2, frame_id_1=i, frame_id_2=i + gap)


def _same_item(a, b, ctx):
    assert set(a) == set(b), (ctx, set(a) ^ set(b))
    for k in a:
        va, vb = a[k], b[k]
        assert type(va) == type(vb), (ctx, k, type(va), type(vb))
        if torch.is_tensor(va):
            assert va.dtype == vb.dtype and va.shape == vb.shape, (ctx, k, va.dtype, vb.dtype, va.shape, vb.shape)
            assert torch.equal(va, vb), (ctx, k)
        elif isinstance(va, np.ndarray):
            assert va.dtype == vb.dtype and np.array_equal(va, vb), (ctx, k)
        else:
            assert va == vb, (ctx, k, va, vb)


@pytest.mark.parametrize('subsample', [False])
def test_pair_pack_reader_matches_the_reference_dataset(tmp_path, monkeypatch, subsample):
    H, W, NF = 16, 24, 6
    root = str(tmp_path)
    _write_tree(root, 'dog', NF, H, W, with_seg=True)
    monkeypatch.chdir(root)                         # the reference hard-codes ./datafiles/davis_processed
    opt = SimpleNamespace(track_id='dog', gaps='1,2', repeat=2, subsample=subsample, overfit=False, cache=False, select=False)
    with ref_exec.on_reference_path():
        import scripts.preprocess.davis.generate_sequence_midas as G      # the reference's WRITER
        out = os.path.join(G.save_path_root, 'dog', '001')
        os.makedirs(out)
        packs = {}
        for gap, bs in ((1, 1), (2, 2)):            # bs = 1 is the shipped value (:179); bs = 2: a multi-pair pack
            for cnt, f in enumerate(np.arange(NF - bs - gap)):
                seq = G.collate_sequence_fix_gap('dog', np.arange(f, f + bs), gap=gap)
                path = os.path.join(out, 'shuffle_False_gap_%02d_sequence_%05d.pt' % (gap, cnt))
                torch.save(seq, path)
                packs[path] = seq
        from datasets.davis_sequence import Dataset as RefDataset
        model = SimpleNamespace(requires=['img', 'flow'], preprocess=lambda s: s)
        ref_items = {}
        for mode in ('train', 'vali'):
            ds = RefDataset(opt, mode=mode, model=model)
            ref_items[mode] = (len(ds), ds.n_frames, [ds[i] for i in range(len(ds))])
    from dvd_hip.datasets.davis_sequence import Dataset, write_pair_pack
    for mode in ('train', 'vali'):
        ds = Dataset(opt, mode=mode, model=model)
        n, nf, items = ref_items[mode]
        assert len(ds) == n and ds.n_frames == nf and n > 0
        for i in range(n):
            _same_item(ds[i], items[i], (mode, i))
    # the product's writer (used by the synthetic-video tests / bench --feed host) emits the reference writer's layout
    some = sorted(packs)[-1]
    ref_pack = packs[some]
    item = Dataset(opt, mode='train', model=model)[len(packs) - 1]
    mine = os.path.join(root, 'mine.pt')
    write_pair_pack(mine, item)
    got = torch.load(mine)
    assert set(got) == set(ref_pack), set(got) ^ set(ref_pack)
    for k in ref_pack:
        assert got[k].shape == ref_pack[k].shape and got[k].dtype == ref_pack[k].dtype, (k, got[k].shape, ref_pack[k].shape)
        if k not in ('depth_1', 'depth_pred_1'):     # the synthetic writer has no MVS / initial depth to store
            assert torch.equal(got[k], ref_pack[k]), k


def test_reference_mask_code_matches_the_oracle_and_the_fixture():
    from oracle import preprocess as OP
    gd = helpers.load_golden('flow_masks')
    for H, W, seed, noise in helpers.FLOW_MASK_CASES:
        f12, f21 = helpers.flow_pair(H, W, seed, noise)
        r1, r2 = ref_exec.reference_masks(f12.numpy(), f21.numpy())
        o1, o2 = OP.consistency_masks(f12.numpy(), f21.numpy())
        assert r1.dtype == np.uint8 and (r1 != o1).sum() == 0 and (r2 != o2).sum() == 0
        assert np.array_equal(np.packbits(r1), gd['mask_1_%dx%d' % (H, W)])
        assert np.array_equal(np.packbits(r2), gd['mask_2_%dx%d' % (H, W)])


def test_checkpoints_interchange_with_the_reference_netinterface(tmp_path):
    from dvd_hip import flat, synthetic
    o = dict(helpers.FULL_STEP_OPT)
    o.update(midas=False, full_logdir=str(tmp_path))
    ck_ref, ck_mine = str(tmp_path / 'ref.pt'), str(tmp_path / 'mine.pt')
    batch = synthetic.make_batch(1, 32, 48, gap=1, seed=9)
    with ref_exec.on_reference_path():
        ref = ref_exec.reference_model(o)
        helpers.seeded_fill_(ref.net_depth, 21)
        helpers.seeded_fill_(ref.net_sceneflow, 22)
        ref.to(torch.device('cpu'))
        ref._train_on_batch(6, 0, helpers.loader_batch(batch))          # non-warm: both Adam optimisers get state
        ref._train_on_batch(6, 1, helpers.loader_batch(batch))
        ref.save_state_dict(ck_ref, save_optimizer=True, additional_values={'epoch': 6})
        ref_sd = [{k: v.clone() for k, v in n.state_dict().items()} for n in ref._nets]
        ref_opt = [op.state_dict() for op in ref._optimizers]
    # ---- reference file -> product
    from dvd_hip.models.scene_flow_motion_field import Model
    with pytest.warns(UserWarning):
        mine = Model(SimpleNamespace(**o), None)
    extra = mine.load_state_dict(ck_ref)
    assert extra == {'epoch': 6}
    for net, sd in zip(mine._nets, ref_sd):
        got = net.state_dict()
        assert list(got) == list(sd)
        for k in sd:
            assert torch.equal(got[k], sd[k]), k
    # the Adam state waits for .to(device) (train.py:256 restores before :279 moves); the flat buffers take it as they would there
    assert mine._pending_optimizer_state is not None and len(mine._pending_optimizer_state) == 2
    flats = [flat.FlatNet(mine.net_depth, o['lr'], (0.5, 0.9)), flat.FlatNet(mine.net_sceneflow, o['lr'] * o['scene_lr_mul'], (0.5, 0.9))]
    for fn, st in zip(flats, mine._pending_optimizer_state):
        fn.load_state_dict(st)
    for fn, st, net in zip(flats, ref_opt, mine._nets):
        assert fn.step_count == 2
        assert len(st['state']) > 0
        for i, p in enumerate(net.parameters()):
            if i not in st['state']:          # never received a gradient (the hourglass' unused uncertainty head)
                assert not fn.view(fn.exp_avg, i).any() and not fn.view(fn.exp_avg_sq, i).any()
                continue
            assert torch.equal(fn.view(fn.exp_avg, i), st['state'][i]['exp_avg'])
            assert torch.equal(fn.view(fn.exp_avg_sq, i), st['state'][i]['exp_avg_sq'])
    # ---- product file -> reference (fresh reference model, different weights)
    mine._optimizers = flats
    mine.save_state_dict(ck_mine, save_optimizer=True, additional_values={'epoch': 7})
    with ref_exec.on_reference_path():
        ref2 = ref_exec.reference_model(o)
        ref2.to(torch.device('cpu'))
        extra = ref2.load_state_dict(ck_mine)
        assert extra == {'epoch': 7}
        for net, sd in zip(ref2._nets, ref_sd):
            for k, v in net.state_dict().items():
                assert torch.equal(v, sd[k]), k
        for op, st in zip(ref2._optimizers, ref_opt):
            got = op.state_dict()
            assert got['param_groups'][0]['lr'] == op.param_groups[0]['lr']        # this run's hyper-parameters kept
            for i in st['state']:
                assert float(got['state'][i]['step']) == 2.0
                assert torch.equal(got['state'][i]['exp_avg'], st['state'][i]['exp_avg'])
                assert torch.equal(got['state'][i]['exp_avg_sq'], st['state'][i]['exp_avg_sq'])
        # and the restored reference model keeps training
        log = ref2._train_on_batch(6, 2, helpers.loader_batch(batch))
        assert np.isfinite(log['loss'])
