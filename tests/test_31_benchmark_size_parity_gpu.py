"""Full-step parity at the BENCHMARK'S image size (384x672; MiDaS non-warm gap 1 = the benchmark, and -- round 4 -- gap 2, the
warm-up phase and the hourglass depth net, i.e. the other schedules of the shipped run): one `_train_on_batch` of the HIP
Model against the CPU oracle's step (oracle/train_step.py, pinned to the real reference's logged losses and gradients by
tests/golden/fullstep_*.npz) on the same frame pair and the same seeded weights -- the comparison `bench.py` reports as
`parity` next to `cpu_baseline`.  One pair: the reference path needs > 60 GB of autograd state per pair at this size.

Tolerances = about 3-5x the values measured on MI355X (round 3, gpurun_out/r03a/parity.jsonl: losses 6e-8 .. 1.9e-7,
acc_reg 3e-9, worst per-parameter gradient norm of the depth net 1.9e-5, worst MLP gradient element 4.2e-5 of its tensor's
largest): losses rtol 2e-6, acc_reg 1e-6, gradient norms 1e-4, MLP elements 2e-4.  The measured values are printed and
written to $DVD_PARITY_LOG (json lines) when set.  Round 4 measured: gap 2 6.9e-8 / 3.4e-5, warm-up 9.1e-8 / -, hourglass 2.4e-7 /
1.6e-5 (loss / worst gradient norm), MLP gradient elements 1.5e-5 .. 7.6e-5.

fp16 ACTIVATION storage (BASELINE configs[4]) against the SAME fp32 oracle: losses rtol 2e-3, gradient norms 5e-2, MLP gradient
elements 2e-2 of max -- the stated tolerance of that mode (tests/test_10_act_fp16_gpu.py has the kernel-level bounds)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


_FIRST = {}


def _oracle_first(gap=1, warm=False, depth='midas'):
    """bench.oracle_first_step, once per configuration (30 s of CPU each: the fp32 and the fp16-activation comparison of the
    benchmark configuration share theirs)."""
    import bench
    key = (gap, warm, depth)
    if key not in _FIRST:
        _FIRST[key] = bench.oracle_first_step(gap=gap, warm=warm, depth=depth)
    return _FIRST[key]


CASES = [
    # gap, warm, depth net, gradient-norm tolerance
    (1, False, 'midas', 1e-4),          # the benchmark configuration (bench.py reports this one as `parity`)
    (2, False, 'midas', 1e-4),          # two Euler steps: BOTH regulariser evaluations shared with the chain (round 4)
    (1, True, 'midas', 1e-4),           # warm-up phase: frozen depth net, L2 criterion, no regulariser
    (1, False, 'hourglass', 1e-4),      # the reference's default depth net (5x5 / 7x7 / 11x11 branches on csrc/xwgrad.hip)
]


@pytest.mark.parametrize('gap,warm,depth,gtol', CASES)
def test_hip_step_matches_the_oracle_at_384x672(gap, warm, depth, gtol):
    import bench
    first = _oracle_first(gap, warm, depth)
    par = bench.hip_parity(first, torch.device('cuda', 0))
    print('parity at 384x672:', json.dumps(par))
    if os.environ.get('DVD_PARITY_LOG'):
        with open(os.environ['DVD_PARITY_LOG'], 'a') as f:
            f.write(json.dumps({'test': 'benchmark_size', 'gap': gap, 'warm': warm, 'depth': depth, **par}) + '\n')
    assert par['rel'] < 2e-6
    for k in ('flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'):
        assert par[k + '_rel'] < 2e-6, k
    if not warm:
        assert par['acc_reg_rel'] < 1e-6
        assert par['depth_grad_norm_worst_rel'] < gtol, par['depth_grad_norm_worst_param']
    assert par['mlp_grad_worst_of_max'] < 2e-4


def test_fp16_activation_step_matches_the_fp32_oracle_at_384x672():
    import bench
    first = _oracle_first()
    par = bench.hip_parity(first, torch.device('cuda', 0), act_fp16=True)
    print('fp16-activation parity at 384x672:', json.dumps(par))
    if os.environ.get('DVD_PARITY_LOG'):
        with open(os.environ['DVD_PARITY_LOG'], 'a') as f:
            f.write(json.dumps({'test': 'benchmark_size_fp16_activations', **par}) + '\n')
    assert not par['step_skipped']
    assert par['rel'] < 2e-3
    for k in ('flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'):
        assert par[k + '_rel'] < 2e-3, k
    assert par['acc_reg_rel'] < 2e-3
    assert par['depth_grad_norm_worst_rel'] < 5e-2, par['depth_grad_norm_worst_param']
    assert par['mlp_grad_worst_of_max'] < 2e-2


def _cfg4_size_parity(reference):
    """One frame pair at BASELINE configs[4]'s own image size (768 x 1344): the fp16-activation HIP step against `reference`
    ('oracle': the fp32 CPU oracle, ~240 GB of host memory and a few minutes of CPU; 'hip': the fp32-storage HIP step)."""
    import bench
    old = bench.H, bench.W
    bench.H, bench.W = 768, 1344
    try:
        first = bench.oracle_first_step() if reference == 'oracle' else bench.hip_fp32_first_step()
        par = bench.hip_parity(first, torch.device('cuda', 0), act_fp16=True)
    finally:
        bench.H, bench.W = old
    par['reference'] = 'CPU oracle (fp32, ATen-CPU), %.0f s for its step' % first['seconds'] if reference == 'oracle' else first['reference']
    print('fp16-activation parity at 768x1344 (vs %s):' % reference, json.dumps(par))
    if os.environ.get('DVD_PARITY_LOG'):
        with open(os.environ['DVD_PARITY_LOG'], 'a') as f:
            f.write(json.dumps({'test': 'configs4_size_fp16_activations_vs_' + reference, **par}) + '\n')
    assert '768x1344' in par['sample']
    assert not par['step_skipped']
    assert par['rel'] < 2e-3
    for k in ('flow_loss_1_2', 'disp_loss_1_2', 'sf_loss'):
        assert par[k + '_rel'] < 2e-3, k
    assert par['acc_reg_rel'] < 2e-3
    assert par['depth_grad_norm_worst_rel'] < 5e-2, par['depth_grad_norm_worst_param']
    assert par['mlp_grad_worst_of_max'] < 2e-2


@pytest.mark.timeout(1500)
def test_fp16_activation_step_at_768x1344_matches_the_cpu_oracle():
    """BASELINE configs[4]'s OWN image size against the ORACLE (the driver-run evidence at this shape was the
    path compared with itself).  The oracle holds ~60 GB of autograd state per pair at 384 x 672, four times that here: the
    test runs where the host has >= 330 GB available (the GPU boxes report 2.9 TB) and skips elsewhere -- the HIP-vs-HIP case
    below always runs.  Same bounds as the fp16 mode's 384 x 672 case against the oracle (measured in round 5 by bench.py
    --config 4: loss 9.7e-7, worst gradient norm 7.1e-4, MLP gradients 1.2e-3 of max)."""
    import bench
    avail = bench.host_mem_available_gb()
    if avail < 330.0:
        pytest.skip('the CPU oracle at 768 x 1344 needs ~240 GB of host memory, %.0f GB are available' % avail)
    _cfg4_size_parity('oracle')


def test_fp16_activation_step_at_768x1344_matches_the_fp32_storage_step():
    """The fall-back that runs on every box: the reference is the fp32-storage HIP step -- what the 384 x 672 cases above pin
    to the oracle (6e-8 / 3e-5).  Same bounds."""
    _cfg4_size_parity('hip')
