"""World-size-2 `gloo` tests (CPU) of the data-parallel exchange of the step
(dvd_hip/parallel.py, dvd_hip/flat.py).

What must hold: pairs are sharded contiguously over ranks; every rank produces the four
UN-normalised loss sums and UN-normalised gradients of its shard; ONE all-reduce(sum) of
the sums and ONE all-reduce(sum) of the flat gradient buffer, followed by the global
1/(sum(mask)+1e-8), reproduce the single-process result on the concatenated batch.  The
per-shard arithmetic here is the CPU oracle (the GPU kernels are checked against the same
oracle in the -m gpu tests), so this file exercises exactly the host-side N>1 logic.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dvd_hip import parallel, synthetic

B, H, W = 4, 16, 24
CAM_KEYS = ('R_1', 'R_2', 'R_1_T', 'R_2_T', 't_1', 't_2', 'K', 'K_inv')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from oracle import sceneflow_mlp as M
    batch = synthetic.make_batch(B, H, W, gap=1, seed=77, with_images=False)
    d1, d2 = synthetic.make_depths(B, H, W, seed=5, far_depth_frac=0.02)
    return batch, d1, d2, M.init_params(seed=3)


def _slice(batch, lo, hi):
    return {k: (v[lo:hi] if (torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B) else v) for k, v in batch.items()}


def _unnormalised(opt, sd, batch, d1, d2):
    """Shard-local sums S0..S3 and gradients of flow_mul*S1 + disp_mul*S2 (what the fused
    kernels emit before the late normalisation)."""
    from oracle import losses as L
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    a = d1.detach().clone().requires_grad_(True)
    b = d2.detach().clone().requires_grad_(True)
    pred = L.predict_train(opt, leaves, batch, a, b)
    _, parts, _ = L.train_losses(opt, False, batch, pred)
    den = parts['mask_sum'] + 1e-8
    S = torch.stack([parts['mask_sum'], parts['flow_loss_1_2'] * den, parts['disp_loss_1_2'] * den,
                     parts['sf_loss'] * den])
    (opt.flow_mul * S[1] + opt.disp_mul * S[2]).backward()
    keys = sorted(leaves)
    flat = torch.cat([leaves[k].grad.reshape(-1) for k in keys])
    return S.detach(), flat, a.grad, b.grad


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_from_env(backend='gloo')
    try:
        from oracle.losses import default_opt
        assert parallel.is_distributed() and parallel.world_size() == world and parallel.rank() == rank
        opt = default_opt()
        batch, d1, d2, sd = _case()
        lo, hi = parallel.shard_range(B)
        S, g, ga, gb = _unnormalised(opt, sd, _slice(batch, lo, hi), d1[lo:hi], d2[lo:hi])
        sums = torch.zeros(8)
        sums[:4] = S
        parallel.all_reduce_sum_(sums)                     # exchange 1: the five loss sums
        inv = 1.0 / (sums[0] + 1e-8)
        h = parallel.all_reduce_sum_async_(g)              # exchange 2: one flat gradient buffer
        h.wait()
        t = torch.tensor([float(rank + 1)])
        parallel.broadcast_(t, 0)
        q.put((rank, lo, hi, sums.numpy().copy(), (g * inv).numpy().copy(), (ga * inv).numpy().copy(),
               (gb * inv).numpy().copy(), float(t)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_every_pair_once():
    for n in (1, 2, 7, 48, 384):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_no_process_group_is_a_noop():
    assert not parallel.is_distributed() and parallel.world_size() == 1 and parallel.rank() == 0
    t = torch.ones(3)
    assert parallel.all_reduce_sum_(t) is t and parallel.all_reduce_sum_async_(t) is None
    assert float(t.sum()) == 3.0


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process_on_concatenated_batch():
    from oracle.losses import default_opt
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    opt = default_opt()
    batch, d1, d2, sd = _case()
    S, g, ga, gb = _unnormalised(opt, sd, batch, d1, d2)
    inv = 1.0 / (S[0] + 1e-8)
    assert [(r[1], r[2]) for r in res] == [(0, 2), (2, 4)]
    for r in res:
        assert r[3][0] == float(S[0])                                     # mask count: exact
        np.testing.assert_allclose(r[3][:4], S.numpy(), rtol=2e-6)
        scale = float((g * inv).abs().max())
        np.testing.assert_allclose(r[4], (g * inv).numpy(), rtol=1e-4, atol=1e-3 * scale)   # weight grads vs ||g||inf
        assert r[7] == 1.0                                                # broadcast from rank 0
    # depth gradients stay shard-local; normalised by the GLOBAL mask sum they tile the full result
    # (the MLP's sgemm blocks differently at B=2 and B=4, so an |.|-loss sign may flip on a near-zero residual)
    for got, want in ((np.concatenate([res[0][5], res[1][5]]), (ga * inv).numpy()),
                      (np.concatenate([res[0][6], res[1][6]]), (gb * inv).numpy())):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-3 * float(np.abs(want).max()))
    # and a per-rank normaliser (what naive DDP averaging would do) is NOT the same thing
    assert abs(res[0][3][0] - 2 * float(_unnormalised(opt, sd, _slice(batch, 0, 2), d1[:2], d2[:2])[0][0])) > 0


def _flat_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend='gloo')
    try:
        from dvd_hip import flat
        torch.manual_seed(rank)                           # ranks start from DIFFERENT weights on purpose
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 1), torch.nn.Conv2d(5, 3, 1))
        fn = flat.FlatNet(net, 1e-3, (0.5, 0.9))
        parallel.broadcast_(fn.flat)                       # train.py:290-292 semantics, one message
        for p in net.parameters():
            assert p.data_ptr() >= fn.flat.data_ptr() and p.grad.data_ptr() >= fn.grad.data_ptr()
        fn.zero_grad()
        net(torch.full((1, 3, 2, 2), float(rank + 1))).sum().backward()
        fn.all_reduce_grads()
        q.put((rank, fn.flat.numpy().copy(), fn.grad.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_buffers_broadcast_and_all_reduce_as_single_messages():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])     # same weights everywhere after the broadcast
    np.testing.assert_array_equal(res[0][2], res[1][2])     # same summed gradient everywhere
    assert np.abs(res[0][2]).sum() > 0
