"""`--dataset davis_sequence`: the per-video pair-pack reader, and the host -> HBM feeder of the step.

Counterpart of /root/reference/datasets/davis_sequence.py:22-154 (`Dataset`): same flags (:25-33), same file
discovery (`<root>/sequences_select_pairs_midas/<track>/001/shuffle_False_gap_%02d_*.pt` per requested gap, frames
counted from `<root>/frames_midas/<track>/*.npz`, :60-84), and the same sample dict per item (:98-115 for
training packs, :117-153 for validation frames).  The `.pt` pack layout is the one
scripts/preprocess/davis/generate_sequence_midas.py:117-179 writes: a dict of tensors concatenated over the
pairs of the pack on dim 0 (`bs = 1` in the shipped script; any `bs` reads here, so a video can be packed as
48-pair files and one file is one optimisation step).

What is new is the transport.  The reference hands CPU tensors to `NetInterface.load_batch`, which copies them
on the compute stream when the step starts.  `DeviceFeeder` keeps two pinned staging buffers per tensor and a
copy stream: while step i runs, pack i+1 is read (DataLoader workers), staged and copied (0.75 GB per 48-pair
step at 384x672 = ~15 ms on PCIe Gen5), so the step never waits for the host.
"""
from glob import glob
from os.path import join

import numpy as np
import torch
import torch.utils.data as data

DATA_ROOT = './datafiles/davis_processed'
FRAME_PREFIX, SEQ_PREFIX = 'frames_midas', 'sequences_select_pairs_midas'


class Dataset(data.Dataset):
    @classmethod
    def add_arguments(cls, parser):
        parser.add_argument('--cache', action='store_true', help='cache the data into ram')
        parser.add_argument('--subsample', action='store_true', help='subsample the video in time')
        parser.add_argument('--track_id', default='train', type=str, help='the track id to load')
        parser.add_argument('--overfit', action='store_true', help='overfit and see if things works')
        parser.add_argument('--gaps', type=str, default='1,2,3,4', help='gaps for sequences')
        parser.add_argument('--repeat', type=int, default=1, help='number of repeatition')
        parser.add_argument('--select', action='store_true', help='pred')
        return parser, set()

    def __init__(self, opt, mode='train', model=None, data_root=None):
        super().__init__()
        assert mode in ('train', 'vali')
        self.opt, self.mode = opt, mode
        root = data_root or getattr(opt, 'data_root', None) or DATA_ROOT
        track = opt.track_id
        if model is None:
            self.required, self.preproc = ['img', 'flow'], None
        else:
            self.required = model.requires if mode == 'train' else ['img']
            self.preproc = model.preprocess
        if mode == 'train':
            sub = 'subsample' if getattr(opt, 'subsample', False) else '%03d' % 1
            path = join(root, SEQ_PREFIX, track, sub)
            self.file_list = []
            for g in (int(x) for x in opt.gaps.split(',')):
                self.file_list += sorted(glob(join(path, 'shuffle_False_gap_%02d_*.pt' % g)))
            self.n_frames = len(glob(join(root, FRAME_PREFIX, track, '*.npz'))) + 0.0
        else:
            self.file_list = sorted(glob(join(root, FRAME_PREFIX, track, '*.npz')))
            self.n_frames = len(self.file_list) + 0.0

    def __len__(self):
        return len(self.file_list) * (self.opt.repeat if self.mode == 'train' else 1)

    def __getitem__(self, idx):
        idx = idx % (self.opt.capat if getattr(self.opt, 'overfit', False) else len(self.file_list))
        unit = 2.0 if getattr(self.opt, 'subsample', False) else 1.0
        s = {}
        if self.mode == 'train':
            pack = torch.load(self.file_list[idx])
            _, H, W, _ = pack['img_1'].shape
            pack['img_1'] = pack['img_1'].permute([0, 3, 1, 2])
            pack['img_2'] = pack['img_2'].permute([0, 3, 1, 2])
            for k, v in pack.items():
                if type(v) != list:
                    s[k] = v.float()
            s['time_step'] = unit / self.n_frames
            for i in ('1', '2'):
                s['time_stamp_' + i] = (pack['fid_' + i].reshape([-1, 1, 1, 1]).expand(-1, -1, H, W) / self.n_frames).float()
                s['frame_id_' + i] = np.asarray(pack['fid_' + i])
        else:
            fr = np.load(self.file_list[idx])
            H, W, _ = fr['img'].shape
            s['time_stamp_1'] = np.ones([1, H, W]) * idx / self.n_frames
            s['img'] = np.transpose(fr['img'], [2, 0, 1])
            s['frame_id_1'] = idx
            s['time_step'] = unit / self.n_frames
            s['depth_pred'] = fr['depth_pred'][None, ...]
            s['depth_mvs'] = fr['depth_mvs'][None, ...]
            s['cam_c2w'] = fr['pose_c2w']
            R, t, K = fr['pose_c2w'][:3, :3], fr['pose_c2w'][:3, 3], fr['intrinsics']
            # stored transposed: row vectors multiply from the left (generate_sequence_midas.py:69-76)
            s['R_1'], s['R_1_T'] = R.T.reshape(1, 1, 3, 3).copy(), R.reshape(1, 1, 3, 3).copy()
            s['t_1'] = t.reshape(1, 1, 1, 3).copy()
            s['K'], s['K_inv'] = K.T.reshape(1, 1, 3, 3).copy(), np.linalg.inv(K).T.reshape(1, 1, 3, 3).copy()
        s['pair_path'] = self.file_list[idx]
        for k, v in s.items():                       # base_dataset.convert_to_float32
            if isinstance(v, np.ndarray):
                s[k] = torch.from_numpy(v).float()
        return s


def write_pair_pack(path, batch):
    """Write a (synthetic) batch in the `.pt` layout generate_sequence_midas.py:156-179 produces, so the reader
    above sees exactly what the reference's preprocessing would have left on disk."""
    B, _, H, W = batch['img_1'].shape
    pack = {k: batch[k].cpu() for k in ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv', 'flow_1_2', 'flow_2_1',
                                        'mask_1', 'mask_2', 'motion_seg_1')}
    pack['img_1'] = batch['img_1'].permute(0, 2, 3, 1).contiguous().cpu()          # stored [B,H,W,3] (:148-149)
    pack['img_2'] = batch['img_2'].permute(0, 2, 3, 1).contiguous().cpu()
    pack['depth_1'] = torch.zeros(B, 1, H, W)
    pack['depth_pred_1'] = torch.ones(B, 1, H, W)
    pack['fid_1'], pack['fid_2'] = batch['frame_id_1'].cpu().float(), batch['frame_id_2'].cpu().float()
    torch.save(pack, path)


class DeviceFeeder(object):
    """Iterates over `loader` (batches of CPU tensors) and yields them resident in HBM, one batch ahead of the
    consumer: pinned double-buffered staging + a dedicated copy stream.  The yielded dict is valid until the next
    `next()`; non-tensor entries pass through."""

    def __init__(self, loader, device, keys=None):
        self.loader, self.device, self.keys = loader, torch.device(device), keys
        self.stream = torch.cuda.Stream(device=self.device)
        self.pinned = [dict(), dict()]
        self.copied = [None, None]       # event behind the last H2D copies out of each slot's pinned buffers

    def _stage(self, batch, slot):
        out, pins = {}, self.pinned[slot]
        if self.copied[slot] is not None:
            # the non-blocking copies issued from this slot two batches ago read the pinned buffers asynchronously: they
            # must have drained before the host overwrites them (a consumer without a per-step host sync would otherwise
            # receive a torn batch)
            self.copied[slot].synchronize()
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if not torch.is_tensor(v) or (self.keys is not None and k not in self.keys) or v.dim() == 0:
                    out[k] = v
                    continue
                buf = pins.get(k)
                if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
                    buf = torch.empty(v.shape, dtype=v.dtype).pin_memory()
                    pins[k] = buf
                buf.copy_(v)
                out[k] = buf.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.copied[slot] = ev
        return out, ev

    def __iter__(self):
        it, slot = iter(self.loader), 0
        try:
            nxt = self._stage(next(it), slot)
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            slot ^= 1
            try:
                nxt = self._stage(next(it), slot)        # copy of the next batch overlaps the consumer's step
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in cur.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(torch.cuda.current_stream(self.device))
            yield cur

    def __len__(self):
        return len(self.loader)
