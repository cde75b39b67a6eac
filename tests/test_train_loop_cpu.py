"""Host-side trainer logic that needs no GPU: `train_epoch` is called exactly as the reference's train.py
calls it (train.py:339-348), the Adam state in checkpoints has the torch.optim layout (netinterface.py:528-562),
a checkpoint restored BEFORE `.to(device)` (train.py:256 then :279) is applied when the flat buffers exist, and
ranks with different local decisions agree on one collective schedule (world size 2, gloo)."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dvd_hip import flat, parallel
from dvd_hip.models.netinterface import NetInterface, NullLogger


class _Toy(NetInterface):
    def __init__(self, opt, logger=None):
        super().__init__(opt, logger)
        self.net = torch.nn.Conv2d(1, 1, 1)
        self._nets = [self.net]
        self._metrics = ['loss']
        self.seen = []

    def _train_on_batch(self, epoch, i, batch):
        self.seen.append(('train', epoch, i, float(batch)))
        return {'size': 1, 'loss': float(batch)}

    def _vali_on_batch(self, epoch, i, batch):
        self.seen.append(('vali', epoch, i, float(batch)))
        return {'size': 1, 'loss': 2 * float(batch)}


class _Recorder(NullLogger):
    def __init__(self):
        super().__init__()
        self.events, self.params, self.model = [], None, None

    def set_params(self, p):
        self.params = p

    def set_model(self, m):
        self.model = m

    def on_epoch_begin(self, e):
        self.events.append(('epoch_begin', e))

    def on_epoch_end(self, e, log=None):
        self.events.append(('epoch_end', e, dict(log)))

    def on_train_end(self, *_a):
        self.events.append(('train_end',))


def _opt():
    return SimpleNamespace(optim='adam', adam_beta1=0.5, adam_beta2=0.9, full_logdir='/tmp')


def test_train_epoch_takes_the_reference_call():
    log = _Recorder()
    m = _Toy(_opt(), log)
    loader = torch.utils.data.DataLoader(torch.arange(5.0), batch_size=1)
    vali = torch.utils.data.DataLoader(torch.arange(3.0), batch_size=1)
    epochs_seen, resets = [], []

    class DS(object):
        def reset(self):
            resets.append(1)

        def __len__(self):
            return 5

    # the exact keyword set of train.py:339-348 plus the two the signature adds (netinterface.py:193-207)
    m.train_epoch(loader, dataloader_vali=vali, max_batches_per_train=4, epochs=2, initial_epoch=3,
                  max_batches_per_vali=2, vali_at_start=True, train_epoch_callback=epochs_seen.append,
                  global_rank=0, reset_dataset=DS())
    assert epochs_seen == [3, 4]                      # DistributedSampler.set_epoch after every training epoch
    assert len(resets) == 2
    assert log.model is m and log.params['steps'] == 4 and log.params['steps_eval'] == 2
    assert log.params['epochs'] == 4 and log.params['metrics'] == ['loss']
    kinds = [(k, e) for k, e, *_ in m.seen]
    assert kinds == ([('vali', 2)] * 2 + [('train', 3)] * 4 + [('vali', 3)] * 2 + [('train', 4)] * 4 + [('vali', 4)] * 2)
    ends = [e for e in log.events if e[0] == 'epoch_end']
    assert ends[1][1] == 3 and ends[1][2]['loss'] == pytest.approx((0 + 1 + 2 + 3) / 4)
    assert log.events[-1] == ('train_end',)
    m._register_tensorboard('tb')
    assert m.tensorboard_logger == 'tb'


def test_missing_batch_log_and_missing_vali_data_raise_like_the_reference():
    m = _Toy(_opt(), None)
    m._train_on_batch = lambda *a: None
    with pytest.raises(ValueError, match='Batch log'):
        m.train_epoch([torch.tensor(1.0)])
    with pytest.raises(ValueError, match='eval_at_beginning'):
        _Toy(_opt(), None).train_epoch([torch.tensor(1.0)], vali_at_start=True)


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))


def test_adam_state_interchanges_with_torch_optim_adam():
    """A torch.optim.Adam checkpoint (what the reference writes) loads into the flat buffers, and what the
    flat buffers write loads into a torch.optim.Adam: same moments, same step."""
    ref_net = _net()
    opt = torch.optim.Adam(ref_net.parameters(), lr=1e-3, betas=(0.5, 0.9))
    for _ in range(3):
        opt.zero_grad()
        ref_net(torch.randn(2, 3, 8, 8)).square().sum().backward()
        opt.step()
    fn = flat.FlatNet(_net(), 5e-4, (0.5, 0.9))
    fn.load_state_dict(opt.state_dict())
    assert fn.step_count == 3 and fn.lr == 5e-4                  # this run's hyper-parameters are kept
    for i, p in enumerate(ref_net.parameters()):
        assert torch.equal(fn.view(fn.exp_avg, i), opt.state[p]['exp_avg'])
        assert torch.equal(fn.view(fn.exp_avg_sq, i), opt.state[p]['exp_avg_sq'])
    # and back: a fresh torch Adam accepts the flat optimiser's state_dict
    net2 = _net()
    opt2 = torch.optim.Adam(net2.parameters(), lr=1e-3, betas=(0.5, 0.9))
    opt2.load_state_dict(fn.state_dict())
    for p, q in zip(net2.parameters(), ref_net.parameters()):
        assert float(opt2.state[p]['step']) == 3.0
        assert torch.equal(opt2.state[p]['exp_avg'], opt.state[q]['exp_avg'])
    # torch 1.9 checkpoints (the reference's pinned version) store `step` as a python int
    old = opt.state_dict()
    for st in old['state'].values():
        st['step'] = int(st['step'])
    fn2 = flat.FlatNet(_net(), 5e-4, (0.5, 0.9))
    fn2.load_state_dict(old)
    assert fn2.step_count == 3
    # never-stepped optimiser (warm-up phase: the depth net has no Adam state yet)
    fn3 = flat.FlatNet(_net(), 5e-4, (0.5, 0.9))
    fn3.load_state_dict(torch.optim.Adam(_net().parameters()).state_dict())
    assert fn3.step_count == 0 and float(fn3.exp_avg.abs().max()) == 0.0
    assert fn3.state_dict()['state'] == {}


def test_checkpoint_restored_before_to_device_is_applied_when_the_buffers_exist(tmp_path):
    """`--resume`: train.py calls load_state_dict(load_optimizer='auto') before model.to(device)."""
    a = _Toy(_opt(), None)
    a._optimizers = [flat.FlatNet(a.net, 1e-3, (0.5, 0.9))]
    a._optimizers[0].exp_avg.fill_(0.25)
    a._optimizers[0].exp_avg_sq.fill_(0.5)
    a._optimizers[0].step_count = 11
    path = str(tmp_path / 'checkpoint.pt')
    a.save_state_dict(path, save_optimizer=True, additional_values={'epoch': 7})
    b = _Toy(_opt(), None)                      # `_optimizers` is still empty, as before Model.to(device)
    extra = b.load_state_dict(path, load_optimizer='auto')
    assert extra == {'epoch': 7} and b._pending_optimizer_state is not None
    b._optimizers = [flat.FlatNet(b.net, 1e-3, (0.5, 0.9))]     # what Model.to(device) does ...
    b._apply_optimizer_state(b._pending_optimizer_state)         # ... followed by this
    fb = b._optimizers[0]
    assert fb.step_count == 11 and float(fb.view(fb.exp_avg, 0).min()) == 0.25 and float(fb.view(fb.exp_avg_sq, 1).max()) == 0.5
    for p, q in zip(a.net.parameters(), b.net.parameters()):
        assert torch.equal(p, q)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _plan_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend='gloo')
    try:
        out = []
        # step 1: rank 1 cannot keep the whole batch's stashes (longer frame gap) and holds a smaller batch
        out.append(parallel.agree_on_step_plan(torch.device('cpu'), rank == 1, 48 if rank == 0 else 40))
        # step 2: both can
        out.append(parallel.agree_on_step_plan(torch.device('cpu'), False, 48))
        # step 3: only rank 0 still has a depth-net graph to capture in phase 3 -> both keep the collective out of flight
        out.append(parallel.agree_on_step_plan(torch.device('cpu'), False, 48, may_capture=(rank == 0)))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_agree_on_one_collective_schedule():
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] == [(True, 88), (False, 96), (False, 96, True)]
    assert parallel.agree_on_step_plan(torch.device('cpu'), False, 5, may_capture=False) == (False, 5, False)
    assert parallel.agree_on_step_plan(torch.device('cpu'), True, 5) == (True, 5)       # no process group: local values


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend='gloo')
    try:
        g = torch.Generator().manual_seed(7 + rank)
        flat = torch.randn(10 * 1024 + 37, generator=g)
        want = flat.clone()
        dist.all_reduce(want)
        pending = parallel.all_reduce_sum_buckets_async_(flat, 4)
        seen = []
        for lo, hi, work in pending:            # consume bucket by bucket, as the optimiser does
            work.wait()
            seen.append((lo, hi, bool(torch.equal(flat[lo:hi], want[lo:hi]))))
        q.put((rank, seen, bool(torch.equal(flat, want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_gradient_all_reduce_equals_one_all_reduce():
    """The depth-net gradient goes out as a few large buckets in flight at once (flat.FlatNet.all_reduce_and_adam_step);
    same sums as one all-reduce, same bucket schedule on every rank."""
    assert parallel.bucket_bounds(10, 4, align=4) == [(0, 4), (4, 8), (8, 10)]
    assert parallel.bucket_bounds(105362945, 4)[-1][1] == 105362945 and len(parallel.bucket_bounds(105362945, 4)) == 4
    assert parallel.bucket_bounds(100, 8) == [(0, 100)]                      # smaller than one aligned bucket
    ctx = mp.get_context('spawn')
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, ok0), (r1, s1, ok1) = sorted(res)
    assert ok0 and ok1 and s0 == s1 and len(s0) == 4 and all(ok for _, _, ok in s0)
    assert s0[0][0] == 0 and s0[-1][1] == 10 * 1024 + 37 and all(a[1] == b[0] for a, b in zip(s0, s0[1:]))
