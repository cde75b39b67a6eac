"""Data parallelism over frame pairs: one process per GPU, RCCL over xGMI.

The path shards by independent units (frame pairs) and has exactly two exchange
steps per optimisation step:
  1. all-reduce(sum) of the 5 loss sums [S0..S3, sum|sf1-sf0|] BEFORE gradients are
     normalised, so that N ranks reproduce the single-device result on the
     concatenated batch (the reference's normaliser is batch-global,
     models/scene_flow_motion_field.py:297-306,340);
  2. all-reduce(sum) of the gradients, as a few large flat buffers (one per
     network): xGMI is point-to-point, so fewer / larger messages are what keeps
     the 7 links busy; 421 MB (MiDaS) is ~5 ms on a ring, ~0.7 ms direct.
The reference wraps its nets in DistributedDataParallel but discards the wrapped
modules (train.py:285-287), so it never synchronises gradients; this module is new
behaviour, not a port.  Everything degrades to a no-op when torch.distributed is not
initialised (single GPU).
"""
import os

import torch
import torch.distributed as dist


# A ONE-rank process group normally takes the single-GPU short cuts below (no collective at all).  With this switch -- the
# DVD_FORCE_DIST=1 environment variable or force_distributed(True) -- a one-rank group runs every collective of the N-rank
# step for real (plan agreement, loss-sum all-reduce, asynchronous MLP-gradient all-reduce under the depth-net backward, the
# bucketed gradient all-reduce pipelined with Adam): on a 1-GPU box that is what executes the RCCL calls of this package,
# stream semantics of `work.wait()` and interaction with captured HIP graphs included (tests/test_33_rccl_one_rank_gpu.py,
# bench.py's `rccl_one_rank` record).  The arithmetic is unchanged: a sum over one rank is the identity.
_FORCE = [os.environ.get('DVD_FORCE_DIST', '0') not in ('', '0')]


def force_distributed(flag=True):
    """Run the collectives of the data-parallel step even in a one-rank group.  Returns the previous setting."""
    old = _FORCE[0]
    _FORCE[0] = bool(flag)
    return old


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE[0])


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def rccl_info():
    """Version of the collective library behind the 'nccl' backend (RCCL on ROCm), for bench lines and test logs."""
    try:
        v = torch.cuda.nccl.version()
        return '.'.join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:          # noqa: BLE001 -- informational only
        return 'unknown (%s)' % type(e).__name__


def init_one_rank(backend='nccl', port=None):
    """A ONE-rank process group on this process's current GPU + force_distributed(): every RCCL call of the N-rank step
    executes on the box at hand.  Returns {'backend', 'rccl_version', 'comm_hbm_bytes'} -- the HBM the communicator took at
    its first collective, measured with hipMemGetInfo around it (what head_room_fraction has to cover per rank)."""
    if dist.is_initialized():
        raise RuntimeError('init_one_rank: a process group exists already')
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    dev = torch.device('cuda', torch.cuda.current_device())
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    dist.init_process_group(backend=backend, init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    warm = torch.zeros(1, device=dev)
    dist.all_reduce(warm)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(dev)[0]
    force_distributed(True)
    return {'backend': backend, 'rccl_version': rccl_info() if backend == 'nccl' else None,
            'comm_hbm_bytes': int(max(0, free0 - free1))}


def shutdown():
    force_distributed(False)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def init_from_env(backend=None):
    """torchrun-style initialisation (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return int(os.environ.get('LOCAL_RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # 'nccl' is RCCL on ROCm
    if backend == 'nccl':
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend=backend)
    if backend == 'nccl':
        # RCCL creates its communicator (channels, staging buffers: a few GB of HBM) at the FIRST collective.  Run one now,
        # so that the step's memory planner (Model._keep_slot reads the free HBM) sees what the collectives will occupy
        # instead of finding out after the kept-activation slots have been sized.
        warm = torch.zeros(1, device=torch.device('cuda', torch.cuda.current_device()))
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    return local


def all_reduce_sum_(t):
    """In-place sum over ranks (no averaging: the loss normaliser is global already)."""
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_sum_async_(t):
    if is_distributed():
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    return None


def bucket_bounds(numel, n_buckets, align=1024):
    """Boundaries of `n_buckets` contiguous, `align`-element aligned slices of a flat buffer.  A function of the
    buffer size only, so every rank issues the same collectives in the same order."""
    n_buckets = max(1, min(int(n_buckets), (numel + align - 1) // align))
    per = -(-numel // n_buckets)
    per = -(-per // align) * align
    bounds = [min(i * per, numel) for i in range(n_buckets + 1)]
    bounds[-1] = numel
    return [(lo, hi) for lo, hi in zip(bounds[:-1], bounds[1:]) if hi > lo]


def all_reduce_sum_buckets_async_(flat, n_buckets):
    """Sum-all-reduce of a flat buffer as a few large messages in flight at once: returns [(lo, hi, work)] in issue
    order.  The caller waits for bucket i and consumes flat[lo:hi] (the optimiser step of that slice) while the
    later buckets are still on the links; xGMI is point-to-point, so the buckets stay large (>= 100 MB for the
    421 MB MiDaS gradient) -- what is bought is the overlap of the reduction with its consumer, not message size."""
    out = []
    for lo, hi in bucket_bounds(flat.numel(), n_buckets):
        work = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True) if is_distributed() else None
        out.append((lo, hi, work))
    return out


def agree_on_step_plan(device, needs_late_normaliser, n_pairs, may_capture=None):
    """Makes the collective schedule of a step independent of rank-local data.

    Whether a rank can keep the whole batch's MLP stashes alive (and so all-reduces the loss sums
    BEFORE its MLP backward) depends on its own batch size and frame gap; whether it still has to capture a depth-net HIP
    graph in phase 3 (and therefore keeps the MLP-gradient all-reduce out of flight until afterwards) depends on its own
    memory situation; ranks that decided differently would issue mismatched collectives.  One small sum all-reduce
    settles both: returns (any rank needs the late normaliser, pairs in the global batch[, any rank may capture])."""
    if not is_distributed():
        out = (bool(needs_late_normaliser), int(n_pairs))
        return out if may_capture is None else out + (bool(may_capture),)
    v = torch.tensor([1.0 if needs_late_normaliser else 0.0, float(n_pairs), 1.0 if may_capture else 0.0], device=device,
                     dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    late, total, cap = v.tolist()
    out = (late > 0.5, int(round(total)))
    return out if may_capture is None else out + (cap > 0.5,)


def broadcast_(t, src=0):
    if is_distributed():
        dist.broadcast(t, src)
    return t


def shard_range(n_items, r=None, w=None):
    """Contiguous shard [lo, hi) of n_items for rank r of w."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n_items, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)
