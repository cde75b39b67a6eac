"""Trainer base with the surface `train.py` and the loggers of the reference expect.

Counterpart of /root/reference/models/netinterface.py:35-601 (`NetInterface`), reduced
to what the test-time-optimisation loop uses: the attribute contract
(`_nets`, `_optimizers`, `_metrics`, `input_names`, `_input`), `load_batch`,
`train_epoch` (per-batch callbacks `on_batch_begin/end`, per-epoch
`on_epoch_begin/end`, netinterface.py:246-360), `to/train/eval/num_parameters`, and the
checkpoint format `{'nets': [...], 'optimizers': [...], **extra}`
(netinterface.py:528-562).  The loggers themselves are out of scope (SURVEY.md section 2
row 13): any object with the callback methods works; `NullLogger` is the default.
"""
import time
from types import SimpleNamespace

import torch

from .. import parallel


class NullLogger(object):
    """Accepts every callback of the reference's ComposeLogger and records the last logs."""

    def __init__(self):
        self.batch_logs, self.epoch_logs = [], []

    def add_logger(self, *_a, **_k):
        pass

    def get_html_logger(self):
        return None

    def set_params(self, *_a, **_k):
        pass

    def set_model(self, *_a, **_k):
        pass

    def train(self):
        pass

    def eval(self):
        pass

    def on_train_begin(self, *_a):
        pass

    def on_train_end(self, *_a):
        pass

    def on_epoch_begin(self, *_a):
        pass

    def on_epoch_end(self, epoch, log=None):
        self.epoch_logs.append((epoch, log))

    def on_batch_begin(self, *_a):
        pass

    def on_batch_end(self, i, log=None):
        self.batch_logs.append(log)


class _EpochMean(object):
    """Size-weighted epoch means of the batch logs (loggers/loggers.py:88-110)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.tot, self.n = {}, 0

    def add(self, log):
        size = log.get('size', 1)
        self.n += size
        for k, v in log.items():
            if k in ('batch', 'size', 'epoch') or not isinstance(v, (int, float)):
                continue
            self.tot[k] = self.tot.get(k, 0.0) + v * size

    def get_epoch_log(self):
        return {k: v / max(self.n, 1) for k, v in self.tot.items()}


class NetInterface(object):
    @classmethod
    def add_arguments(cls, parser):
        return parser, set()

    @staticmethod
    def preprocess(sample_loaded):
        return sample_loaded

    def __init__(self, opt, logger=None):
        self._logger = logger if logger is not None else NullLogger()
        self._internal_logger = _EpochMean()
        self.opt = opt
        self.full_logdir = getattr(opt, 'full_logdir', None)
        if opt.optim != 'adam':
            raise NotImplementedError('the fused HIP step implements Adam (the shipped --optim); got %s' % opt.optim)
        self.optim_params = {'betas': (opt.adam_beta1, opt.adam_beta2)}
        self._nets, self._optimizers, self._metrics = [], [], []
        self._moveable_vars = []
        self.input_names, self.gt_names, self.aux_names = [], [], []
        self._input, self._gt, self._aux = SimpleNamespace(), SimpleNamespace(), SimpleNamespace()
        self.device = torch.device('cpu')
        self._pending_optimizer_state = None

    def init_vars(self, add_path=True):
        for name in self.input_names:
            setattr(self._input, name, None)
        for name in self.gt_names:
            setattr(self._gt, name, None)

    def init_weight(self, net=None, init_type='kaiming', init_param=0.02, a=0):
        """Same policy as netinterface.py:55-86 for conv / linear / batch-norm modules."""
        from torch.nn import init

        def fn(m):
            cname = m.__class__.__name__
            if hasattr(m, 'weight') and m.weight is not None and ('Conv' in cname or 'Linear' in cname):
                if init_type == 'normal':
                    init.normal_(m.weight.data, 0.0, init_param)
                elif init_type == 'xavier':
                    init.xavier_normal_(m.weight.data, gain=init_param)
                elif init_type == 'kaiming':
                    init.kaiming_normal_(m.weight.data, a=a, mode='fan_in')
                elif init_type == 'orth':
                    init.orthogonal_(m.weight.data, gain=init_param)
                else:
                    raise NotImplementedError(init_type)
                if getattr(m, 'bias', None) is not None:
                    init.constant_(m.bias.data, 0.0)
            elif 'BatchNorm' in cname and m.affine:
                init.normal_(m.weight.data, 1.0, init_param)
                init.constant_(m.bias.data, 0.0)
        net.apply(fn)

    # -- data movement -------------------------------------------------------------------
    def load_batch(self, batch, include_gt=True):
        for name in self.input_names:
            if name in batch:
                v = batch[name]
                if torch.is_tensor(v) and v.device != self.device:
                    v = v.to(self.device, non_blocking=True)
                setattr(self._input, name, v)
        if include_gt:
            for name in self.gt_names:
                if name in batch:
                    setattr(self._gt, name, batch[name].to(self.device, non_blocking=True))

    def to(self, device):
        for net in self._nets:
            net.to(device)
        self.device = torch.device(device)

    def train(self):
        for m in self._nets:
            m.train()

    def eval(self):
        for m in self._nets:
            m.eval()

    def num_parameters(self, return_list=False):
        counts = [sum(p.numel() for p in net.parameters()) for net in self._nets]
        return counts if return_list else sum(counts)

    # -- checkpoints ---------------------------------------------------------------------
    def save_state_dict(self, filepath, *, save_optimizer=False, additional_values={}):
        sd = {'nets': [net.state_dict() for net in self._nets]}
        if save_optimizer:
            sd['optimizers'] = [o.state_dict() for o in self._optimizers]
        sd.update(additional_values)
        torch.save(sd, filepath)

    def load_state_dict(self, filepath, *, load_optimizer='auto'):
        sd = torch.load(filepath, map_location='cpu')
        if load_optimizer == 'auto':
            load_optimizer = 'optimizers' in sd
        assert len(self._nets) == len(sd['nets'])
        for net, s in zip(self._nets, sd['nets']):
            net.load_state_dict(s)
        if load_optimizer:
            if not self._optimizers:
                # train.py:256 restores the checkpoint BEFORE model.to(device) (:279); the flat Adam buffers
                # only exist on the device, so the state is held here and applied by to()
                self._pending_optimizer_state = list(sd['optimizers'])
            else:
                self._apply_optimizer_state(sd['optimizers'])
        return {k: v for k, v in sd.items() if k not in ('nets', 'optimizers')}

    def _apply_optimizer_state(self, states):
        assert len(self._optimizers) == len(states)
        for o, s in zip(self._optimizers, states):
            o.load_state_dict(s)       # keeps this run's lr / betas / eps (netinterface.py:565-574)

    # -- loop ----------------------------------------------------------------------------
    def _train_on_batch(self, epoch, batch_ind, batch):
        raise NotImplementedError

    def _vali_on_batch(self, epoch, batch_ind, batch):
        raise NotImplementedError

    def test_on_batch(self, batch_ind, batch):
        raise NotImplementedError

    def _register_tensorboard(self, tblogger):
        self.tensorboard_logger = tblogger

    def train_epoch(self, dataloader, *, dataloader_vali=None, max_batches_per_train=None, max_batches_per_vali=None,
                    epochs=1, initial_epoch=1, verbose=1, reset_dataset=None, vali_at_start=False, global_rank=0,
                    train_epoch_callback=None):
        """Signature and callback order of netinterface.py:193-360: per epoch `reset_dataset.reset()`,
        `on_epoch_begin`, per batch `on_batch_begin` / `_train_on_batch` / `on_batch_end`, `on_epoch_end` with the
        (rank-averaged) epoch means, then `train_epoch_callback(epoch)` (train.py passes
        `DistributedSampler.set_epoch`, :308-310,339-348), then validation."""
        logger = self._logger

        def n_samples(loader):
            """_get_num_samples (netinterface.py:26-32): samples the sampler yields, whole batches only with drop_last;
            loaders without a batch sampler (plain iterables, the device feeder over one): one sample per step."""
            bs = getattr(loader, 'batch_sampler', None) or getattr(getattr(loader, 'loader', None), 'batch_sampler', None)
            try:
                if bs is None:
                    return len(loader), 1
                n = len(bs.sampler)
                return (n // bs.batch_size * bs.batch_size if bs.drop_last else n), bs.batch_size
            except TypeError:
                return None, 1

        def limits():
            """steps / samples per training epoch and per validation pass (netinterface.py:217-232), including the
            reference's quirk of clamping the validation samples with the TRAINING loader's batch size."""
            st = len(dataloader) if hasattr(dataloader, '__len__') else None
            sa, bsz = n_samples(dataloader) if st is not None else (None, 1)
            if max_batches_per_train is not None:
                st = max_batches_per_train if st is None else min(st, max_batches_per_train)
                if sa is not None:
                    sa = min(sa, st * bsz)
            sv = sav = 0
            if dataloader_vali is not None:
                sv = len(dataloader_vali)
                sav, _ = n_samples(dataloader_vali)
                if max_batches_per_vali is not None:
                    sv = min(sv, max_batches_per_vali)
                    if sav is not None:
                        sav = min(sav, sv * bsz)
            return st, sa, sv, sav

        def announce():
            st, sa, sv, sav = limits()
            logger.set_params({'epochs': epochs + initial_epoch - 1, 'steps': st, 'steps_eval': sv, 'samples': sa,
                               'samples_eval': sav, 'verbose': 1, 'metrics': self._metrics})      # 'verbose': 1 (:239)
            return st, sv

        steps, steps_eval = announce()
        logger.set_model(self)
        logger.on_train_begin()
        dataset_size = [0]

        def epoch_log():
            elog = self._internal_logger.get_epoch_log()
            if parallel.is_distributed():
                for k in sorted(elog):
                    v = torch.tensor(elog[k], device=self.device, dtype=torch.float64)
                    parallel.all_reduce_sum_(v)
                    elog[k] = float(v) / parallel.world_size()
            return elog

        def run(epoch, loader, on_batch, limit, is_train):
            (self.train if is_train else self.eval)()
            (logger.train if is_train else logger.eval)()
            self._internal_logger.reset()
            logger.on_epoch_begin(epoch)
            t0 = time.time()
            for i, data in enumerate(loader):
                if limit is not None and i >= limit:
                    break
                start = time.time()
                data_time = start - t0
                logger.on_batch_begin(i)
                log = on_batch(epoch, i, data)
                if log is None:
                    raise ValueError('Batch log is not returned by _train_on_batch method. Aborting.')
                log.update(batch=i, epoch=epoch, data_time=data_time, batch_time=time.time() - start)
                self._internal_logger.add(log)
                logger.on_batch_end(i, log)
                t0 = time.time()
            elog = epoch_log()
            logger.on_epoch_end(epoch, elog)
            return elog

        last = None
        if vali_at_start:
            if dataloader_vali is None:
                raise ValueError('eval_at_beginning is set to True but no eval data is given.')
            with torch.no_grad():
                run(initial_epoch - 1, dataloader_vali, self._vali_on_batch, steps_eval, False)
        for epoch in range(initial_epoch, initial_epoch + epochs):
            if reset_dataset is not None:
                (reset_dataset.reset if hasattr(reset_dataset, 'reset') else reset_dataset)()
                if hasattr(reset_dataset, '__len__') and dataset_size[0] != len(reset_dataset):
                    steps, steps_eval = announce()
                    dataset_size[0] = len(reset_dataset)
            last = run(epoch, dataloader, self._train_on_batch, steps, True)
            if train_epoch_callback is not None:
                train_epoch_callback(epoch)
            if dataloader_vali is not None:
                with torch.no_grad():
                    run(epoch, dataloader_vali, self._vali_on_batch, steps_eval, False)
        logger.on_train_end()
        return last
