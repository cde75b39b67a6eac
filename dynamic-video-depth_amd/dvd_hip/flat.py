"""Flat parameter / gradient / Adam-state buffers for one network.

Laid out for the MI355X step: every parameter of a net becomes a view into ONE
fp32 buffer (16-byte aligned segments), its `.grad` a view into a second one, and
the Adam moments live in two more.  That turns the optimiser into a single fused
HIP launch per net (dvd_adam_step) and the data-parallel exchange into a single
RCCL all-reduce per net instead of one message per tensor (the reference
broadcasts ~600 tensors one by one, train.py:290-292).
"""
import torch

from . import ops, parallel


class FlatNet(object):
    def __init__(self, module, lr, betas, eps=1e-8):
        self.module = module
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        params = [p for p in module.parameters()]
        self.params = params
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4        # keep every segment 16-byte aligned
        self.offsets, self.numel = offs, n
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(params, offs):
            seg = self.flat[o:o + p.numel()].view_as(p)
            seg.copy_(p.data)
            p.data = seg
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.step_count = 0
        # 1-element device view of "steps whose update was skipped so far" (the loss-scale state's count that the guarded
        # Adam kernel subtracts from `step`, csrc/elementwise.hip) or None.  step_count advances on every call, skipped or
        # not -- the host never learns of a skip inside the step -- so the EFFECTIVE step, the one torch.optim.Adam /
        # GradScaler mean and a checkpoint must carry, is step_count - skipped.
        self.skip_count = None

    def _skipped(self):
        return int(self.skip_count.item()) if self.skip_count is not None else 0

    def view(self, buf, p_index):
        p, o = self.params[p_index], self.offsets[p_index]
        return buf[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-attach
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    # -- gradient accumulation over several backward passes (depth-net chunks) ----------------
    # With `.grad` pre-attached autograd adds every parameter's gradient with its own tiny kernel
    # (~620 launches per MiDaS backward).  Instead: detach the views, let the engine hand over the
    # fresh gradient tensors, and add them all into the flat buffer with one multi-tensor call.
    def detach_grads(self):
        for p in self.params:
            p.grad = None

    def reattach_grads(self):
        """`.grad` of every parameter = its view of the flat buffer again (after a detach that was not absorbed)."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def absorb_grads(self):
        views, fresh = [], []
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            v = self.grad[o:o + p.numel()].view_as(p)
            if g is not None:
                views.append(v)
                fresh.append(g if g.is_contiguous() else g.contiguous())
            p.grad = v
        if views:
            torch._foreach_add_(views, fresh)

    def all_reduce_grads(self, async_op=False):
        if async_op:
            return parallel.all_reduce_sum_async_(self.grad)
        parallel.all_reduce_sum_(self.grad)
        return None

    def all_reduce_and_adam_step(self, n_buckets=4, skip_ptr=None):
        """Gradient exchange pipelined with the optimiser: the flat gradient goes out as `n_buckets` large
        all-reduces issued back to back; the Adam kernel of bucket i runs as soon as that bucket has arrived, while
        buckets i+1.. are still being reduced (RCCL runs them on its own stream; `work.wait()` only makes the compute
        stream wait, the host never blocks).  Single process: one Adam launch over the whole buffer."""
        if not parallel.is_distributed():
            return self.adam_step(skip_ptr=skip_ptr)
        pending = parallel.all_reduce_sum_buckets_async_(self.grad, n_buckets)
        self.step_count += 1
        for lo, hi, work in pending:
            if work is not None:
                work.wait()
            ops.adam_step(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.step_count,
                          self.lr, self.betas[0], self.betas[1], self.eps, skip_ptr=skip_ptr)

    def adam_step(self, extra_grad=None, scale=1.0, scale_ptr=None, skip_ptr=None):
        """param -= Adam(scale * (*scale_ptr) * grad + extra_grad); nothing happens if *skip_ptr != 0."""
        self.step_count += 1
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas[0],
                      self.betas[1], self.eps, scale=scale, scale_ptr=scale_ptr, grad2=extra_grad, skip_ptr=skip_ptr)

    # Checkpoints carry `optimizer.state_dict()` of a torch.optim.Adam (netinterface.py:528-536, written by
    # ModelSaveLogger with save_optimizer=True); emit and accept exactly that layout -- per-parameter
    # `step` / `exp_avg` / `exp_avg_sq` keyed by the parameter's index in `module.parameters()` order, one
    # param group -- so checkpoints written by either implementation resume in the other.
    def state_dict(self):
        state = {}
        eff = self.step_count - self._skipped()      # skipped steps did not advance the optimiser
        if eff > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {'step': torch.tensor(float(eff)),
                            'exp_avg': self.exp_avg[o:o + n].view_as(p).clone(),
                            'exp_avg_sq': self.exp_avg_sq[o:o + n].view_as(p).clone()}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts a torch.optim.Adam state_dict (any torch version: `step` int or tensor) and keeps this
        run's lr / betas / eps (optimizer_load_state_dict(keep_training_params=True), netinterface.py:565-574).
        Parameters without state (never stepped) keep zero moments."""
        if 'state' not in sd:                      # round-1 flat layout
            self.step_count = int(sd['step']) + self._skipped()
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            return
        groups = sd['param_groups']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(self.params):
            raise ValueError('optimizer state has %d parameters, the network %d' % (len(order), len(self.params)))
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for slot, key in enumerate(order):
            st = sd['state'].get(key)
            if not st:
                continue
            p, o = self.params[slot], self.offsets[slot]
            n = p.numel()
            if st['exp_avg'].numel() != n:
                raise ValueError('optimizer state of parameter %d has %d elements, expected %d'
                                 % (slot, st['exp_avg'].numel(), n))
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter Adam step counts differ (%s): the fused optimiser keeps one count per '
                             'network' % sorted(steps))
        # (the kernel subtracts this process's skip count: add it, so that the effective step continues from the checkpoint's)
        self.step_count = (steps.pop() if steps else 0) + self._skipped()
