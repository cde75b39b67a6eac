"""More than one step: K = 5 consecutive `_train_on_batch` calls of the HIP Model on ONE batch against the series logged by
the REAL reference's `Model._train_on_batch` (tests/golden/make_golden.py::case_trajectory, fixtures traj5_*.npz): the losses
of every step, and the parameters after the fifth.

Every fixture / parity test before round 5 was ONE step (missing 3 / weak 2).  What K steps add: Adam's
moments and bias correction across steps (flat.py + dvd_adam_step), replayed depth-net graphs following the weights, the
regulariser's shared evaluations on updated weights -- and the DIRECTION of the loss at every step.  (On the MiDaS case at the
one-step fixtures' lr = 1e-4 the reference's OWN loss goes up at the second step, 5.818 -> 7.444 -- a 1e-4 Adam step on a head
calibrated for random weights -- and the product's does the same, 7.4446: the smoke test's "loss changed" assertion is all one
can ask of two steps there.  That configuration is not a fixture: the reference does not reproduce itself on it beyond the
second step, see CASES.)

Tolerances: step 0 is the one-step bound (1e-5).  Later steps see parameters that moved by +-lr per element in the direction
of sign(g): an element whose gradient is within rounding of 0 can move the other way, so the bound grows with the step --
the values measured on MI355X are logged to $DVD_PARITY_LOG and the bounds below are about 3x those.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu

KEYS = ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg')


def _build(gd, **over):
    from dvd_hip import synthetic
    from dvd_hip.models.scene_flow_motion_field import Model
    o = dict(helpers.FULL_STEP_OPT)
    o.update(midas=bool(gd['midas']), full_logdir='/tmp')
    if 'over_keys' in gd:                     # option overrides the fixture was generated with
        o.update({str(k): (bool(v) if isinstance(o.get(str(k)), bool) else float(v)) for k, v in zip(gd['over_keys'], gd['over_vals'])})
    o.update(over)
    opt = SimpleNamespace(**o)
    with pytest.warns(UserWarning):
        model = Model(opt, None)
    seed = int(gd['seed'])
    helpers.seeded_fill_(model.net_depth, seed)
    helpers.seeded_fill_(model.net_sceneflow, seed + 1)
    if opt.midas:
        with torch.no_grad():
            model.net_depth.scratch.output_conv[4].weight.mul_(30.0)
            model.net_depth.scratch.output_conv[4].bias.fill_(2000.0)
    model.to(torch.device('cuda'))
    batch = synthetic.make_batch(int(gd['B']), int(gd['H']), int(gd['W']), gap=int(gd['gap']), seed=seed + 2)
    return model, opt, batch


def _run(gd, **over):
    model, opt, batch = _build(gd, **over)
    series = {k: [] for k in KEYS}
    for i in range(int(gd['steps'])):
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        log = model._train_on_batch(int(gd['epoch']), i, helpers.loader_batch(b))
        for k in KEYS:
            series[k].append(float(log[k]))
    torch.cuda.synchronize()
    return model, opt, series


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


# (fixture, per-step bound on the relative difference of the logged losses, bound on acc_reg, bound on parameter norms)
CASES = [
    # measured on MI355X (profiles/r05_parity_measured.jsonl): 2.0e-7, 1.1e-7, 7.3e-7, 9.8e-7, 1.1e-5; acc_reg <= 3.6e-5
    ('traj5_hourglass_b2_32x48', (1e-5, 1e-5, 1e-5, 2e-5, 5e-5), 2e-4, 1e-4),
    # MiDaS at the shipped learning rates (lr 1e-6, MLP x 1000).  At the one-step fixtures' lr = 1e-4 this case is chaotic in the
    # REFERENCE itself (two CPU runs of the real reference with 2 and 8 threads: 2 % apart after the third step, its loss
    # jumping 5.82 -> 7.44 -> 5.6) -- see make_golden.py; at the shipped rates those two runs agree to 1e-6
    # measured: 4.1e-7, 2.3e-6, 1.9e-5, 2.0e-5, 1.3e-4; acc_reg <= 1.6e-4
    ('traj5_midas_b1_64x96', (1e-5, 2e-5, 1e-4, 1e-4, 5e-4), 1e-3, 1e-4),
]


@pytest.mark.parametrize('name,step_tol,acc_tol,norm_tol', CASES)
def test_five_steps_follow_the_reference(name, step_tol, acc_tol, norm_tol):
    gd = helpers.load_golden(name)
    model, opt, series = _run(gd)
    measured = {'test': 'trajectory/' + name}
    for i in range(int(gd['steps'])):
        worst = max(_rel(series[k][i], float(gd['series_' + k][i])) for k in ('loss', 'flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'))
        measured['step%d_loss_rel' % i] = worst
        measured['step%d_acc_reg_rel' % i] = _rel(series['acc_reg'][i], float(gd['series_acc_reg'][i]))
    print('measured trajectory:', measured, 'loss series', series['loss'], 'reference', gd['series_loss'].tolist())
    if os.environ.get('DVD_PARITY_LOG'):
        import json
        with open(os.environ['DVD_PARITY_LOG'], 'a') as f:
            f.write(json.dumps(measured) + '\n')
    # the direction of every step of the loss is the reference's
    ref = gd['series_loss']
    for i in range(1, len(ref)):
        assert (series['loss'][i] - series['loss'][i - 1]) * (ref[i] - ref[i - 1]) > 0, 'step %d moves the other way' % i
    for i in range(int(gd['steps'])):
        assert measured['step%d_loss_rel' % i] <= step_tol[i], 'step %d: %.3e' % (i, measured['step%d_loss_rel' % i])
        assert measured['step%d_acc_reg_rel' % i] <= (5e-6 if i == 0 else acc_tol), 'acc_reg, step %d' % i
    # parameters after step K: norms, and selected tensors element by element (an element moves by <= lr per step)
    names = [str(n) for n in gd['param_names']]
    want_p = dict(zip(names, gd['param_norms_after']))
    K = int(gd['steps'])
    for prefix, net in (('depth', model.net_depth), ('sf', model.net_sceneflow)):
        lr = opt.lr * (opt.scene_lr_mul if prefix == 'sf' else 1.0)
        for k, p in net.named_parameters():
            key = prefix + '/' + k
            assert abs(float(p.data.double().norm()) - want_p[key]) <= 2 * K * lr * p.numel() ** 0.5 + norm_tol * want_p[key], key
    worst_elem = 0.0
    for k in [k for k in gd if k.startswith('p_sf/') or k.startswith('p_depth/')]:
        prefix, pname = k.split('/', 1)
        net = model.net_sceneflow if prefix == 'p_sf' else model.net_depth
        lr = opt.lr * (opt.scene_lr_mul if prefix == 'p_sf' else 1.0)
        p = dict(net.named_parameters())[pname]
        d = np.abs(p.data.cpu().numpy() - gd[k])
        worst_elem = max(worst_elem, float(d.max() / lr))
        # no element can be further than 2 K lr from the reference's (both moved by <= K lr); most are much closer
        assert d.max() <= 2 * K * lr + 1e-7, k
        assert (d > 0.5 * lr).mean() <= 0.05, '%s: %.1f %% of the elements more than lr/2 away' % (k, 100 * (d > 0.5 * lr).mean())
    helpers.log_measured('trajectory/%s/param_elem_worst_in_lr' % name, worst_elem, 2 * K)


def test_five_steps_fp16_activations_stay_near_the_fp32_reference():
    """`--act_fp16` (BASELINE configs[4]'s arithmetic) over five steps against the REAL reference's fp32 series: the mode's
    per-step gradient error is 1.4-1.7e-2 in the worst parameter-gradient norm (tests/test_10), what that does to a
    trajectory is stated here.  No skipped step, loss scale on the device."""
    gd = helpers.load_golden('traj5_midas_b1_64x96')
    model, opt, series = _run(gd, act_fp16=True)
    st = model._gscale.tolist()
    assert st[5] == 0, 'fp16 overflow guard skipped %d steps' % st[5]
    ref = gd['series_loss']
    rels = [_rel(series['loss'][i], float(ref[i])) for i in range(len(ref))]
    print('fp16 trajectory: loss', series['loss'], 'reference', ref.tolist(), 'rel', rels)
    for i, r in enumerate(rels):
        helpers.log_measured('trajectory/fp16/step%d_loss_rel' % i, r, 2e-4 if i == 0 else 3e-3)
    assert rels[0] <= 2e-4             # measured 1.5e-5
    assert max(rels) <= 3e-3           # measured 5.5e-4 (third step)
    for i in range(1, len(ref)):
        assert (series['loss'][i] - series['loss'][i - 1]) * (ref[i] - ref[i - 1]) > 0, 'step %d moves the other way' % i
