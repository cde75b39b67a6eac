#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X dynamic-video-depth step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --config 4 [--pairs 64]        BASELINE configs[4] on one GPU (768x1344, fp16 activations) with its parity leg
    DVD_RESERVE_GB=24 python bench.py ...           the same with 24 GB of the device taken first (stand-in for RCCL's buffers)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no torch.distributed environment the script re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); DVD_DIST_BACKEND=gloo lets the N ranks
share one GPU (a smoke test of the N > 1 path on a 1-GPU box).

Workload (BASELINE.json configs[1] = configs[2]): synthetic 384x672 video, 48 frame pairs per GPU,
MiDaS depth net (ResNeXt-101 32x8d, random init + calibrated head; every convolution, BatchNorm+ReLU, pooling and
up-sampling on the hand-written HIP kernels of dvd_hip/csrc -- fp16-pair-split MFMA implicit GEMMs with fp32 accumulation)
+ scene-flow MLP, non-warm phase (L1 + acceleration regulariser), gap 1, fp32 storage.  A "step" is one
`Model._train_on_batch`: depth nets forward, geometry + MLP + fused warp/loss forward and
backward, depth-net backward, gradient all-reduce (N>1) and both Adam updates.  Inputs are
resident in HBM before the timed region.

Prints ONE JSON line: whole-job `value` in 48-pair iterations per second (weak scaling:
every rank owns 48 pairs), plus
  roofline     -- the fused warp+loss op (HBM bound), timed live with events on its stream;
  roofline_mfma -- the matrix kernels (85 % of the step): algorithmic work the timed steps executed, counted by the library per
                  kernel class, over the step time, as a fraction of the 2.5 PFLOP/s dense fp16 MFMA peak; per-class kernel
                  times of the three largest classes from the committed trace (provenance stated);
  cpu_baseline -- the CPU oracle (a port of the reference step) on a bounded sample;
  parity       -- the HIP Model against that oracle on the SAME frame pair and weights at the benchmark's image size
                  (losses and gradients of one step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
os.environ.setdefault('MIOPEN_LOG_LEVEL', '1')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

H, W, PAIRS, GAP = 384, 672, 48, 1
PAIRS_CFG4 = 24                  # BASELINE configs[4] (768x1344, fp16 activations): frame pairs per GPU whose depth-net state stays resident (2 x 59 GB of kept slots + 119 GB of MLP stashes); more pairs run with recomputed chunks
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
WARP_BYTES_PER_PIXEL = 52        # fused fwd+bwd, unique bytes
MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (2.5 PFLOP/s; the 5 PF figure is 2:1 sparsity)
# MFMAs issued per algorithmic multiply-accumulate: fp32 storage = two-term fp16 split of both operands,
# three partial products; fp16 activation storage = one-term activations: 2 in forward / backward-data, 1 in the weight
# gradients; the scene-flow MLP keeps three (its weight gradient two with the fp16 stash)
PRODUCTS_FP32 = {k: 3.0 for k in ('xconv_1x1_wide', 'xconv_wide', 'xconv_128', 'xconv_small', 'xwgrad3', 'xwgrad3g', 'xwgrad1b',
                                   'xwgrad1s', 'xwgradk', 'mlp_fwd', 'mlp_bwd_dx', 'mlp_bwd_dw')}
PRODUCTS_FP16 = dict(PRODUCTS_FP32, xconv_1x1_wide=2.0, xconv_wide=2.0, xconv_128=2.0, xconv_small=2.0, xwgrad3=1.0, xwgrad3g=1.0,
                     xwgrad1b=1.0, xwgrad1s=1.0, mlp_bwd_dw=2.0)


def make_opt(**over):
    from types import SimpleNamespace
    o = dict(optim='adam', adam_beta1=0.5, adam_beta2=0.9, lr=1e-6, scene_lr_mul=1000.0, dataset='davis_sequence',
             batch_size=1, global_rank=0, use_cnn=False, use_embedding=False, midas=True, use_disp=True,
             use_disp_ratio=False, time_dependent=True, flow_mul=1.0, disp_mul=1.0, acc_mul=1.0, sf_mag_div=100.0,
             interp_steps=5, warm_reg=False, weight_steps=False, use_motion_seg=False, n_freq_xyz=16, n_freq_t=16,
             warm_sf=5, mlp_stash_gb=64.0, depth_chunk=16, full_logdir='/tmp', act_fp16=False)
    o.update(over)
    return SimpleNamespace(**o)


def build_model(opt, device, seed=0, to_device=True):
    import warnings
    from dvd_hip.models.scene_flow_motion_field import Model
    from dvd_hip.third_party.MiDaS import calibrate_head_for_random_init
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = Model(opt, None)
    if opt.midas:
        calibrate_head_for_random_init(model.net_depth)
    if to_device:
        model.to(device)
    return model


class WarpTimer(object):
    """Brackets every dvd_warp_loss_fused call with events on the launch stream."""

    def __init__(self):
        from dvd_hip import ops
        self.ops, self.orig, self.events, self.pixels, self.on = ops, ops.warp_loss_fused, [], 0, False

    def __enter__(self):
        def wrapped(cfg, *a, **k):
            if not self.on:
                return self.orig(cfg, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(cfg, *a, **k)
            e1.record()
            self.events.append((e0, e1, cfg.B * cfg.H * cfg.W))
            return r
        self.ops.warp_loss_fused = wrapped
        return self

    def __exit__(self, *a):
        self.ops.warp_loss_fused = self.orig

    def summary(self):
        if not self.events:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.events)
        px = sum(n for _, _, n in self.events)
        return {'launches': len(self.events), 'avg_ms': ms / len(self.events), 'pixels_per_launch': px / len(self.events),
                'GBps': px * WARP_BYTES_PER_PIXEL / ms / 1e6}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def oracle_first_step(pairs=1, gap=None, warm=False, depth='midas'):
    """The oracle's first optimisation step (seeded weights) on `pairs` frame pairs of the benchmark workload at 384x672:
    returns everything the cpu_baseline leg continues from and the parity leg compares with -- the initial weights,
    the batch, the step's log and the gradients its Adam steps consumed.  gap / warm / depth: the other schedules of the
    shipped run at the same image size (frame gap = Euler steps, the warm-up phase, the hourglass depth net)."""
    from dvd_hip import synthetic
    from dvd_hip.third_party.hourglass import HourglassModel_Embed
    from dvd_hip.third_party.MiDaS import MidasNet, calibrate_head_for_random_init
    from oracle import sceneflow_mlp as M
    from oracle import train_step as T
    from oracle.losses import default_opt
    gap = GAP if gap is None else gap
    torch.manual_seed(0)
    if depth == 'midas':
        net = calibrate_head_for_random_init(MidasNet(non_negative=True, normalize_input=True)).eval()
    else:
        net = HourglassModel_Embed(noexp=False, use_embedding=False).eval()
    sd = M.init_params(seed=0)
    init = ({k: v.detach().clone() for k, v in net.state_dict().items()}, {k: v.clone() for k, v in sd.items()})
    opt = default_opt(midas=depth == 'midas')
    batch = synthetic.make_batch(pairs, H, W, gap=gap, seed=1234)
    state = {}
    t0 = time.time()
    log, tm = T.train_step(opt, net, sd, batch, warm=warm, lr_depth=1e-6, lr_mlp=1e-3, adam_state=state, return_grads=True)
    return {'net': net, 'sd': sd, 'opt': opt, 'batch': batch, 'state': state, 'init': init, 'log': log, 'gap': gap, 'warm': warm,
            'depth': depth, 'mlp_grads': tm['mlp_grads'], 'depth_grad_norms': tm['depth_grad_norms'],
            'seconds': time.time() - t0}


def hip_parity(first, device, act_fp16=False):
    """One `_train_on_batch` of the HIP Model on the oracle's pair and initial weights; returns the `parity` object of
    the bench line: both losses, relative differences of every logged loss, the worst relative difference of the
    per-parameter gradient norms of the depth net and the worst element of the MLP gradients (relative to each
    tensor's largest element)."""
    from dvd_hip import synthetic
    opt = make_opt(depth_chunk=1, midas=first.get('depth', 'midas') == 'midas', act_fp16=bool(act_fp16))
    model = build_model(opt, torch.device('cpu'), seed=0, to_device=False)
    model.net_depth.load_state_dict(first['init'][0])
    model.net_sceneflow.load_state_dict(first['init'][1])
    model.to(device)
    b = {k: (v.to(device) if torch.is_tensor(v) and k != 'time_step' else v) for k, v in first['batch'].items()}
    warm = bool(first.get('warm', False))
    log = model._train_on_batch(opt.warm_sf + (0 if warm else 1), 0, synthetic.with_loader_dim(b))
    torch.cuda.synchronize()
    rel = lambda a, c: abs(a - c) / max(abs(c), 1e-30)       # noqa: E731
    out = {'sample': '%d frame pair(s) at %dx%d, gap %d, %s depth net, %s phase, seeded weights, one step' % (
               first['batch']['img_1'].shape[0], H, W, first.get('gap', GAP), first.get('depth', 'midas'), 'warm-up' if warm else 'main'),
           'loss_cpu': first['log']['loss'], 'loss_hip': log['loss'], 'rel': rel(log['loss'], first['log']['loss'])}
    for k in ('flow_loss_1_2', 'disp_loss_1_2', 'sf_loss', 'acc_reg'):
        out[k + '_rel'] = rel(log[k], first['log'][k])
    worst, worst_name = 0.0, None
    for k, p in model.net_depth.named_parameters():
        want = first['depth_grad_norms'].get(k)
        if not want or p.grad is None:
            continue
        r = rel(float(p.grad.double().norm()), want)
        if r > worst:
            worst, worst_name = r, k
    out['depth_grad_norm_worst_rel'], out['depth_grad_norm_worst_param'] = worst, worst_name
    wm = 0.0
    for k, p in model.net_sceneflow.named_parameters():
        g = first['mlp_grads'][k]
        wm = max(wm, float((p.grad.cpu() - g).abs().max() / g.abs().max().clamp_min(1e-30)))
    out['mlp_grad_worst_of_max'] = wm
    if act_fp16:
        st = model._gscale.tolist()
        out['activations'] = 'fp16'
        out['loss_scale_log2'] = __import__('math').log2(st[0]) if st[0] > 0 else None
        out['step_skipped'] = bool(st[4])
    del model
    return out


def hip_fp32_first_step(pairs=1, gap=None):
    """The fp32-storage HIP Model's first step on the benchmark workload at the CURRENT image size, in the shape
    `oracle_first_step` returns: the reference of the fp16-activation parity leg where the CPU oracle does not fit the host
    (BASELINE configs[4]: 768 x 1344 needs ~4x the 60 GB the oracle holds at 384 x 672)."""
    from dvd_hip import synthetic
    gap = GAP if gap is None else gap
    device = torch.device('cuda', torch.cuda.current_device())
    opt = make_opt(depth_chunk=1, midas=True, act_fp16=False)
    model = build_model(opt, torch.device('cpu'), seed=0, to_device=False)
    init = ({k: v.detach().clone() for k, v in model.net_depth.state_dict().items()},
            {k: v.detach().clone() for k, v in model.net_sceneflow.state_dict().items()})
    model.to(device)
    batch = synthetic.make_batch(pairs, H, W, gap=gap, seed=1234)
    b = {k: (v.to(device) if torch.is_tensor(v) and k != 'time_step' else v) for k, v in batch.items()}
    t0 = time.time()
    log = model._train_on_batch(opt.warm_sf + 1, 0, synthetic.with_loader_dim(b))
    torch.cuda.synchronize()
    out = {'init': init, 'batch': batch, 'log': {k: float(v) for k, v in log.items()}, 'gap': gap, 'warm': False, 'depth': 'midas',
           'mlp_grads': {k: p.grad.detach().cpu().clone() for k, p in model.net_sceneflow.named_parameters()},
           'depth_grad_norms': {k: float(p.grad.double().norm()) for k, p in model.net_depth.named_parameters() if p.grad is not None},
           'seconds': time.time() - t0, 'reference': 'HIP Model with fp32 activation storage (the arithmetic the 384 x 672 parity legs '
                                                   'check against the CPU oracle)'}
    del model
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def host_mem_available_gb():
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                return float(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def cpu_baseline(first, timed_steps=3, budget_s=240.0):
    """The oracle step (a port of the reference's `_train_on_batch`, pinned to the real reference's logs by
    tests/golden/fullstep_*.npz) on the host cores: ONE frame pair of the same workload at 384x672 -- the
    48-pair step needs >400 GB of autograd state on the CPU path -- 1 warm-up step (`first`) +
    `timed_steps` timed steps, median; value = pairs/s / 48.  Stops timing early
    once `budget_s` of CPU time is spent (at least one timed step)."""
    from oracle import train_step as T
    threads = torch.get_num_threads()
    times, log, warm_s = [], first['log'], first['seconds']
    t_all = time.time()
    for i in range(timed_steps):
        t0 = time.time()
        log, _ = T.train_step(first['opt'], first['net'], first['sd'], first['batch'], warm=False, lr_depth=1e-6, lr_mlp=1e-3,
                              adam_state=first['state'])
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {'value': (1.0 / med) / PAIRS, 'unit': 'iters/s (48-pair steps)', 'cores': threads, 'kind': 'port',
            'sample': '1 frame pair at %dx%d, gap %d: 1 warm-up step (%.1f s) + %d timed oracle steps, median %.1f s '
                      '(all: %s); value = pairs/s / 48' % (H, W, GAP, warm_s, len(times), med,
                                                           ', '.join('%.1f' % t for t in times)),
            'pairs_per_s': 1.0 / med, 'first_step_loss': first['log']['loss'], 'cpu_model': _cpu_model(),
            'os_cpu_count': os.cpu_count(), 'torch_threads': threads}


def _valu_issue(pixels_per_launch):
    """The warp+loss tile kernel's OTHER roofline: vector-instruction issue.  Static, from the committed SQ counters of the
    CURRENT kernel (profiles/warp_loss_sq.json, written by tools/sq_to_json.py from a rocprofv3 --pmc pass of the
    micro-benchmark at 48 x 384 x 672): wave instructions x 64 lanes / pixels, and the time they take at 100 % issue."""
    path = os.path.join(ROOT, 'profiles', 'warp_loss_sq.json')
    try:
        rec = json.load(open(path))
        ipp = rec['valu_wave_instructions'] * 64.0 / rec['pixels']
        from dvd_hip import build as _build
        return {'instr_per_pixel': round(ipp, 1), 'issue_bound_ms': round(ipp * pixels_per_launch / (256 * 64 * rec['clock_hz']) * 1e3, 4),
                'kernel': rec.get('kernel'), 'source': 'static: profiles/warp_loss_sq.json (%s)' % rec.get('collected', ''),
                'profile_is_of_this_binary': _profile_is_current(rec, _build.WARP_UNITS)}
    except Exception:                      # noqa: BLE001 -- informational only
        return None


def _roofline_mfma(f0, f1, steps, dt, world, act_fp16, r0=None, r1=None):
    """Matrix-pipe roofline of the step (the kernels that own ~85 % of it): the algorithmic work the timed steps EXECUTED per
    kernel class -- counted by the library at launch / graph-capture time and at every graph replay (dvd_flop_counters,
    ops.executed_flops) -- over the timed wall clock, as fp32-equivalent TFLOP/s, as issued fp16 MFMA TFLOP/s (x products per
    multiply-accumulate of the arithmetic in use) and as a fraction of the 2.5 PFLOP/s dense peak.  Per-class kernel TIME is
    not measurable in this process; the committed trace of the same command supplies it (profiles/mfma_roofline.json, written by
    tools/mfma_roofline.py; provenance stated) for the three classes with the most time."""
    prod = PRODUCTS_FP16 if act_fp16 else PRODUCTS_FP32
    per_step = {k: (f1[k] - f0[k]) / steps for k in f1 if k in prod}
    total = sum(per_step.values())
    issued = sum(per_step[k] * prod[k] for k in per_step)
    s_per_step = dt / steps
    out = {'bound': 'mfma', 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'algorithmic_TFLOP_per_step': total / 1e12,
           'algorithmic_TFLOP_per_step_by_class': {k: round(v / 1e12, 4) for k, v in per_step.items() if v},
           'achieved_fp32_equivalent': total / s_per_step / 1e12,
           'achieved': issued / s_per_step / 1e12, 'frac': issued / s_per_step / 1e12 / MFMA_PEAK_TFLOPS,
           'products_per_mac': {k: prod[k] for k in per_step if per_step[k]},
           'note': 'step-wide: all matrix-kernel work of a step over the WHOLE step time (per GPU; the other ~15 % of the step '
                   'is memory-bound helper kernels); achieved = issued fp16 MFMA rate = sum(work x products per MAC) / time'}
    if r0 is not None and r1 is not None:
        # the part of that work that is RE-computation (ops.RECOMPUTED: depth-net forwards of chunks whose autograd state was not
        # kept, the stash-free Euler chain of the MLP's recompute schedule) is executed, not needed
        rec_step = {k: (r1[k] - r0[k]) / steps for k in per_step}
        needed = {k: per_step[k] - rec_step[k] for k in per_step}
        out['recomputed_TFLOP_per_step'] = sum(rec_step.values()) / 1e12
        out['needed_TFLOP_per_step'] = sum(needed.values()) / 1e12
        out['frac_of_needed_work'] = sum(needed[k] * prod[k] for k in needed) / s_per_step / 1e12 / MFMA_PEAK_TFLOPS
    path = os.path.join(ROOT, 'profiles', 'mfma_roofline.json')
    try:
        rec = json.load(open(path))
        key = 'fp16' if act_fp16 else 'fp32'
        top = []
        for r in rec.get(key, {}).get('classes', [])[:3]:
            w = per_step.get(r['class'], 0.0)
            ms = r['ms_per_step']
            top.append({'class': r['class'], 'kernels': r['kernels'], 'ms_per_step': ms, 'share_of_step_kernel_time': r.get('share'),
                        'algorithmic_TFLOP_per_step': w / 1e12, 'fp32_equivalent_TFLOPs': w / (ms * 1e-3) / 1e12 if ms else None,
                        'issued_TFLOPs': w * prod[r['class']] / (ms * 1e-3) / 1e12 if ms else None,
                        'frac_of_peak': w * prod[r['class']] / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if ms else None})
        out['top_kernels'] = top
        out['top_kernels_source'] = 'static kernel times: profiles/mfma_roofline.json (%s); work: counted live in this run' % \
            rec.get(key, {}).get('collected', '')
        out['profile_is_of_this_binary'] = _profile_is_current(rec.get(key, {}), None)
    except Exception:                      # noqa: BLE001 -- informational only
        out['top_kernels'] = None
    return out


def _profile_is_current(rec, units):
    """Was a committed profile that this line joins collected from the kernels this process runs?  The tools that write
    profiles/*.json on the GPU box stamp them with dvd_hip.build.source_digest (sources + flags of the translation units
    concerned); True / False, or None for a profile from before the stamp existed."""
    try:
        from dvd_hip import build
        want = rec.get('source_digest')
        return None if not want else bool(want == build.source_digest(units))
    except Exception:                      # noqa: BLE001 -- informational only
        return None


def _roofline_helpers(f0, f1, steps, act_fp16):
    """The memory-bound helper kernels of a step (BatchNorm+ReLU, up-sampling, max|.| scalars, weight packing, pooling, the
    stage-1 grouped convolution on the vector unit, element-wise passes, Adam, un-projection): algorithmic bytes per step per
    class, counted live by the library (dvd_byte_counters: every operand read once, every result written once), over the
    class's kernel time per step from the committed trace (profiles/mfma_roofline.json `helpers`, written by
    tools/mfma_roofline.py from the same two traces as the matrix classes; static, provenance and digest stated) = GB/s against
    the 8 TB/s HBM peak.  `other_ms_per_step` is what neither a matrix nor a helper class claims (ATen / runtime kernels)."""
    from dvd_hip import ops
    per_step = {k: (f1[k] - f0[k]) / steps for k in ops.BYTE_CLASSES}
    out = {'unit': 'GB/s', 'peak': HBM_PEAK_GBPS, 'algorithmic_MB_per_step': {k: round(v / 1e6, 2) for k, v in per_step.items() if v}}
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'mfma_roofline.json'))).get('fp16' if act_fp16 else 'fp32', {})
        rows, tot = [], 0.0
        for r in rec.get('helpers', []):
            b, ms = per_step.get(r['class'], 0.0), r['ms_per_step']
            tot += ms
            rows.append({'class': r['class'], 'ms_per_step': ms, 'launches_per_step': r.get('launches_per_step'),
                         'algorithmic_MB_per_step': b / 1e6, 'GBps': b / (ms * 1e-3) / 1e9 if ms else None,
                         'frac': b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms else None})
        out['classes'] = rows
        out['helper_ms_per_step'] = tot
        out['other_ms_per_step'] = rec.get('other_ms_per_step')
        out['other_kernels'] = rec.get('other_kernels')
        out['source'] = 'static kernel times: profiles/mfma_roofline.json (%s); bytes: counted live in this run' % rec.get('collected', '')
        out['profile_is_of_this_binary'] = _profile_is_current(rec, None)
    except Exception:                      # noqa: BLE001 -- informational only
        out['classes'] = None
    return out


def _sub_bench(args, timeout_s, keep):
    """One more bench line from a child process of this script (its own model, its own process group): the keys in `keep` of
    its JSON line, or {'error': ...} -- a failing extra never takes the headline line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + args + ['--no_extras', '--no_cpu_baseline']
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.split('\n') if l.startswith('{')]
        if r.returncode != 0 or not line:
            return {'error': 'exit %d: %s' % (r.returncode, (r.stderr or '')[-300:]), 'command': ' '.join(cmd[1:])}
        rec = json.loads(line[-1])
        out = {k: rec.get(k) for k in keep if k in rec}
        out['command'] = 'python ' + ' '.join(os.path.basename(c) if c.endswith('bench.py') else c for c in cmd[1:])
        return out
    except subprocess.TimeoutExpired:
        return {'error': 'timeout after %d s' % timeout_s, 'command': ' '.join(cmd[1:])}
    except Exception as e:                 # noqa: BLE001 -- informational only
        return {'error': repr(e)}


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a torch.distributed environment: start N ranks of this script."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--pairs', type=int, default=PAIRS, help='pairs per GPU (48 = the BASELINE configuration)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--gap', type=int, default=GAP,
                    help='frame gap of the synthetic pairs = Euler steps of the scene-flow integration (1 = the BASELINE '
                         'configuration; the shipped DAVIS schedule mixes 1..4: extra bench lines, not the headline)')
    ap.add_argument('--depth', choices=('midas', 'hourglass'), default='midas',
                    help="depth network: midas (BASELINE configs[1]/[2]) or the reference's default hourglass (extra bench line)")
    ap.add_argument('--cpu_steps', type=int, default=2,
                    help='timed oracle steps of the cpu_baseline leg (after 1 warm-up; ~50 s each on a 128-thread host: two keep the '
                         'default run, which since round 6 also carries the configs4 and rccl_one_rank child lines, inside ~6 minutes)')
    ap.add_argument('--depth_graphs', type=int, default=int(os.environ.get('DVD_DEPTH_GRAPHS', '1')),
                    help='1 (default): replay the depth net from HIP graphs; 0: eager launches')
    ap.add_argument('--depth_chunk', type=int, default=0,
                    help='images per depth-net forward/backward chunk = per kept-activation graph slot.  0 (default): the model '
                         'chooses at its first step (Model --depth_chunk 0: the largest of 48 / 24 / 16 for which every slot fits beside '
                         'the MLP stashes, else 16) -- 48 on the headline line (two slots, both kept), 16 for the hourglass (six of six '
                         '26 GB slots kept; one of two 77 GB ones at 48: 0.845 -> 0.929 iters/s), at frame gap 2 and for --config 4 above '
                         '24 pairs; the line reports it as depth_chunk')
    ap.add_argument('--act_fp16', action='store_true',
                    help='fp16 ACTIVATION storage in the depth net (fp32 parameters / accumulation / loss sums): the arithmetic of '
                         'BASELINE configs[4] at the headline image size -- an extra bench line, not the headline')
    ap.add_argument('--config', type=int, default=2, choices=(2, 4),
                    help='2 (default): BASELINE configs[1]/[2], the headline.  4: BASELINE configs[4] on ONE GPU -- synthetic '
                         '768x1344, fp16 activations with fp32 loss accumulation, as many frame pairs as fit one MI355X '
                         '(--pairs, default %d)' % PAIRS_CFG4)
    ap.add_argument('--cfg4_parity', choices=('auto', 'oracle', 'hip', 'none'), default='auto',
                    help='--config 4: the parity leg at 768x1344 (one pair): against the fp32 CPU oracle if the host has the memory '
                         '(auto / oracle), else against the fp32-storage HIP step (hip)')
    ap.add_argument('--feed', choices=('hbm', 'host'), default='hbm',
                    help="hbm (default, the contract's `value`): inputs resident in HBM before the timed region; host: every "
                         "step's batch starts in host memory and goes through the pinned double-buffered feeder "
                         "(dvd_hip.datasets.davis_sequence.DeviceFeeder), so the PCIe copy is inside the timed region")
    ap.add_argument('--mlp_recompute', type=int, default=1, choices=(0, 1),
                    help='A/B of the scene-flow MLP schedule when the stashes of the whole batch do not fit (configs[4] at 64 pairs): '
                         '1 = recompute schedule (round 6), 0 = late normaliser (rounds 1-5); see Model.add_arguments')
    ap.add_argument('--depth_keep_gb', type=float, default=None,
                    help='HBM budget of the kept depth-net activation slots (Model.add_arguments: 150); every slot is still '
                         'subject to the free-memory test of keep_slot_fits')
    ap.add_argument('--no_extras', action='store_true',
                    help='only the headline measurement: without the configs4 and rccl_one_rank sub-records the default N = 1 run '
                         'adds from child processes of this script')
    ap.add_argument('--rccl_one_rank', action='store_true',
                    help='N = 1 with a ONE-rank nccl (= RCCL) process group and the collectives of the N-rank step forced '
                         '(dvd_hip.parallel.init_one_rank): plan agreement, loss-sum all-reduces, MLP-gradient all-reduce under the '
                         'depth-net backward, bucketed depth-net gradient all-reduce pipelined with Adam -- the RCCL calls of the '
                         'data-parallel step on a 1-GPU box (the arithmetic is unchanged: a sum over one rank)')
    a = ap.parse_args()
    global H, W
    if a.config == 4:
        H, W = 768, 1344
        a.act_fp16 = True
        if a.pairs == PAIRS:
            a.pairs = PAIRS_CFG4
        if a.pairs > 24 and a.depth_keep_gb is None:
            a.depth_keep_gb = 160.0        # four 38 GB slots of 16 images (the model's default budget, 150 GB, stops at three)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(a.gpus)

    from dvd_hip import parallel, synthetic
    # DVD_DIST_BACKEND=gloo lets two ranks share one GPU (smoke test of the N>1 path on a 1-GPU box);
    # the driver's multi-GPU runs use the default, nccl = RCCL over xGMI, one rank per GPU
    local = parallel.init_from_env(backend=os.environ.get('DVD_DIST_BACKEND'))
    local = local % max(torch.cuda.device_count(), 1)
    rccl_one = None
    if a.rccl_one_rank:
        if a.gpus != 1 or parallel.is_distributed():
            raise SystemExit('--rccl_one_rank is a 1-GPU, 1-process run')
        torch.cuda.set_device(local)
        rccl_one = parallel.init_one_rank('nccl')
    world, rank = parallel.world_size(), parallel.rank()
    if a.gpus != world:
        raise SystemExit('--gpus %d but the torch.distributed world has %d ranks' % (a.gpus, world))
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    if os.environ.get('DVD_CUDNN_BENCHMARK'):
        torch.backends.cudnn.benchmark = True     # MIOpen find mode (experiments; the default run uses FAST immediate mode)
    # DVD_RESERVE_GB=N: take N GB of the device with a raw hipMalloc BEFORE the model's memory planner runs -- a stand-in for what
    # RCCL's communicator holds on an 8-rank run (the planner sees less free memory and keeps fewer depth-net slots); outside
    # torch's allocator, so hbm_peak_reserved_GB below is the model's own
    ballast = None
    if os.environ.get('DVD_RESERVE_GB'):
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        ballast = ctypes.c_void_p()
        nbytes = int(float(os.environ['DVD_RESERVE_GB']) * 2 ** 30)
        if hip.hipMalloc(ctypes.byref(ballast), ctypes.c_size_t(nbytes)) != 0:
            raise SystemExit('DVD_RESERVE_GB: hipMalloc of %d bytes failed' % nbytes)
    opt = make_opt(global_rank=rank, depth_chunk=min(a.depth_chunk, a.pairs), depth_graphs=bool(a.depth_graphs),     # (0 = auto)
                   midas=a.depth == 'midas', act_fp16=bool(a.act_fp16), mlp_recompute=int(a.mlp_recompute))
    if a.depth_keep_gb is not None:
        opt.depth_keep_gb = float(a.depth_keep_gb)
    model = build_model(opt, device, seed=0)
    batch = synthetic.make_batch(a.pairs, H, W, gap=a.gap, seed=1234, rank=rank, device=device)
    epoch = opt.warm_sf + 1            # non-warm phase

    if a.feed == 'host':
        from dvd_hip.datasets.davis_sequence import DeviceFeeder
        host = [synthetic.with_loader_dim({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()})
                for _ in range(2)]

    def one_step(i):
        if a.feed == 'host':
            return model._train_on_batch(epoch, i, dict(next(feeder)))
        return model._train_on_batch(epoch, i, synthetic.with_loader_dim(batch))

    # HIP-graph set-up (like a compile step, outside the W + K steps of the contract): the first step captures the depth
    # net's forward and forward+backward graphs, the second pays their first-launch upload (3.1 s instead of 1.5 s)
    setup_steps = 2 if a.depth_graphs else 0
    if a.feed == 'host':
        feeder = iter(DeviceFeeder((host[i & 1] for i in range(setup_steps + a.warmup + a.steps)), device))
    for i in range(setup_steps):
        one_step(i)
    with WarpTimer() as wt:
        for i in range(a.warmup):
            log = one_step(i)
        if parallel.is_distributed():
            torch.distributed.barrier()
        torch.cuda.synchronize()
        wt.on = True
        from dvd_hip import ops as _ops
        flops0, rec0 = _ops.executed_flops(), dict(_ops.RECOMPUTED)
        t0 = time.time()
        for i in range(a.steps):
            log = one_step(a.warmup + i)
        torch.cuda.synchronize()
        flops1, rec1 = _ops.executed_flops(), dict(_ops.RECOMPUTED)
        if parallel.is_distributed():
            torch.distributed.barrier()
        dt = time.time() - t0
        warp = wt.summary()
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if parallel.is_distributed():
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)
    ms_per_step = dt / a.steps * 1e3
    dist_on = parallel.is_distributed()
    dist_backend = torch.distributed.get_backend() if dist_on else None
    try:
        rccl_version = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                      # noqa: BLE001 -- informational only
        rccl_version = None
    from dvd_hip.models.scene_flow_motion_field import head_room_fraction
    head_room_gb = head_room_fraction(world) * torch.cuda.get_device_properties(device).total_memory / 2 ** 30
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    out = {
        'metric': 'train iters/s (depth+sceneflow step) at %dx%d, %d pairs; warp+loss HBM GB/s' % (H, W, a.pairs if a.config == 4 else PAIRS),
        'value': world * (1.0 if a.config == 4 else a.pairs / float(PAIRS)) * a.steps / dt,
        'unit': ('iters/s (%d-pair steps at %dx%d, whole job)' % (a.pairs, H, W)) if a.config == 4 else 'iters/s (48-pair steps, whole job)',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': ('f16 activations / f32 accumulate (depth-net activations and their gradients stored as fp16, one-term MFMA operands: '
                  '2 MFMAs per activation x weight product, 1 per weight-gradient product; fp32 parameters, parameter gradients, '
                  'loss sums and optimiser state; power-of-two loss scale kept on the device, csrc/a16.hip)') if a.act_fp16 else
                 'f32 (storage and accumulation fp32; conv / MLP contractions on v_mfma_f32_32x32x16_f16: every fp32 operand, '
                 'scaled by a power of two from its tensor\'s max, is split into two fp16 terms (22 bits) and a product is '
                 'three partial products; <= 4e-6 of max|y| against float64, the bound the 3-term bf16 / 6-product '
                 'arithmetic of round 2 met)', 'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[4] on one GPU' if a.config == 4 else
                                ('BASELINE configs[1]/[2] with fp16 activation storage' if a.act_fp16 else 'BASELINE configs[1]/[2]')) +
                               ': synthetic %dx%d, %d frame pairs per GPU, gap %d, MiDaS '
                               '(ResNeXt-101 32x8d) depth net with hand-written fp16-pair-split MFMA convolution kernels '
                               '(forward, data and weight gradients) '
                               'under PyTorch-ROCm autograd + HIP scene-flow MLP + HIP fused warp/reprojection/loss, '
                               'non-warm phase with acceleration regulariser' % (H, W, a.pairs, a.gap),
                   'pairs_per_gpu': a.pairs, 'height': H, 'width': W, 'gap': a.gap, 'depth_net': a.depth,
                   'activations': 'fp16' if a.act_fp16 else 'fp32',
                   'parallelism': 'dp%d over frame pairs' % world},
        'pairs_per_s': world * a.pairs * a.steps / dt, 'feed': a.feed, 'graph_setup_steps': setup_steps,
        'hbm_peak_allocated_GB': torch.cuda.max_memory_allocated(device) / 2 ** 30,
        'hbm_peak_reserved_GB': torch.cuda.max_memory_reserved(device) / 2 ** 30,
        'hbm_graph_pools_GB': getattr(model, '_pool_bytes', 0) / 2 ** 30,      # kept depth-net activations + graph temporaries
        'depth_chunk': model._chunk(),                                          # images per kept slot (--depth_chunk 0: the model's choice)
        'depth_slots_kept': '%d of %d' % (sum(1 for k, v in model._depth_graphs.items() if k[0] == 'keep' and v is not None),
                                          sum(1 for k in model._depth_graphs if k[0] == 'keep')),
        'last_loss': log['loss'],
        # BatchNorm+ReLU sites of the depth net's captured backward passes: how many found their ReLU mask already applied by
        # the consuming convolution's epilogue (conv._Site) and how many ran their own mask pass
        'bn_relu_sites': dict(__import__('dvd_hip.conv', fromlist=['STATS']).STATS),
        'dist_backend': dist_backend, 'ranks_seen': world,
        # the collective library behind torch.distributed's 'nccl' backend on ROCm (RCCL) and the planner's head room
        'rccl_version': rccl_version, 'hbm_head_room_GB': head_room_gb,
        'hbm_reserved_by_others_GB': float(os.environ.get('DVD_RESERVE_GB') or 0.0),      # the DVD_RESERVE_GB ballast
    }
    if warp is not None:
        # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 --pmc passes of
        # the same launch at 48 x 384 x 672: tools/gpu_visit.sh, stage pmc -> tools/pmc_to_json.py); the committed summary of the
        # CURRENT kernel is read here and its provenance is stated next to the number
        traffic, traffic_src, traffic_cur = None, None, None
        pmc = os.path.join(ROOT, 'profiles', 'warp_loss_pmc.json')
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                # (the committed counters are for ONE launch of `algorithmic_bytes_per_launch` / 52 pixels: scale by this
                #  run's pixels per launch -- round 4 scaled by 48 * H * W after --config 4 had changed H and W)
                pmc_pixels = rec.get('algorithmic_bytes_per_launch') / float(WARP_BYTES_PER_PIXEL)
                traffic = rec.get('hbm_bytes_per_launch') * warp['pixels_per_launch'] / pmc_pixels
                traffic_src = 'static: profiles/warp_loss_pmc.json (%s)' % rec.get('collected', 'rocprofv3 --pmc passes')
                from dvd_hip import build as _build
                traffic_cur = _profile_is_current(rec, _build.WARP_UNITS)
            except Exception:
                traffic = None
        out['roofline'] = {'bound': 'hbm', 'kernel': 'dvd_warp_loss_fused (strip kernel + unit combine + reductions; round 6)',
                           'achieved': warp['GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                           'frac': warp['GBps'] / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': traffic_src,
                           'profile_is_of_this_binary': traffic_cur,
                           'algorithmic_bytes_per_launch': warp['pixels_per_launch'] * WARP_BYTES_PER_PIXEL,
                           'avg_launch_ms': warp['avg_ms'], 'launches_timed': warp['launches'],
                           # the kernel's OTHER roofline: it is VALU-issue bound (the reference's exact fp32 rounding sequence
                           # for everything that decides a mask or a tap index).  Static, from the committed SQ counters of
                           # this kernel: wave instructions x 64 lanes / pixels, and the time they take at 100 % issue
                           # (256 CUs x 64 lanes per clock at the measured 2.16 GHz)
                           'valu_issue': _valu_issue(warp['pixels_per_launch'])}
    out['roofline_mfma'] = _roofline_mfma(flops0, flops1, a.steps, dt, world, a.act_fp16, rec0, rec1)
    out['roofline_helpers'] = _roofline_helpers(flops0, flops1, a.steps, a.act_fp16)
    if rccl_one is not None:
        out['rccl_one_rank'] = dict(rccl_one, comm_hbm_GB=rccl_one['comm_hbm_bytes'] / 2 ** 30, collectives_forced=True)
    if a.act_fp16:
        st = model._gscale.tolist()
        out['loss_scale'] = {'log2_S': __import__('math').log2(st[0]) if st[0] > 0 else None, 'target_exponent': st[2],
                             'steps_skipped': st[5]}
    if world == 1 and a.config == 4 and a.cfg4_parity != 'none' and a.depth == 'midas' and a.gap == GAP:
        # BASELINE configs[4]'s own image size: one frame pair, the fp16-activation step against the fp32 CPU oracle where
        # the host holds it (~4 x the 60 GB of 384 x 672; --cfg4_parity oracle forces it), else against the fp32-storage HIP
        # step -- which one is stated in parity.reference
        import gc
        del model, batch
        gc.collect()
        torch.cuda.empty_cache()
        use_oracle = a.cfg4_parity == 'oracle' or (a.cfg4_parity == 'auto' and not a.no_cpu_baseline and host_mem_available_gb() >= 330.0)
        first = oracle_first_step() if use_oracle else hip_fp32_first_step()
        out['parity'] = hip_parity(first, device, act_fp16=True)
        out['parity']['reference'] = ('CPU oracle (fp32, ATen-CPU), %.0f s for its step' % first['seconds']) if use_oracle else \
            first['reference']
        out['parity']['host_mem_available_GB'] = host_mem_available_gb()
    if world == 1 and not a.no_cpu_baseline and a.depth == 'midas' and a.gap == GAP and a.config == 2:
        # the 48-pair model's graph slots hold most of the HBM: release them before the 1-pair parity model is built
        import gc
        del model, batch
        gc.collect()
        torch.cuda.empty_cache()
        first = oracle_first_step()
        out['parity'] = hip_parity(first, device, act_fp16=a.act_fp16)      # (fp16 activations: against the SAME fp32 oracle)
        out['cpu_baseline'] = cpu_baseline(first, timed_steps=max(1, a.cpu_steps))
    if world == 1 and not a.no_extras and a.config == 2 and not a.act_fp16 and a.depth == 'midas' and a.gap == GAP \
            and a.feed == 'hbm' and not a.rccl_one_rank and a.pairs == PAIRS:
        # Two more lines of the SAME script, from child processes, inside the headline line (so that the driver's BENCH record
        # carries them): BASELINE configs[4] at its own 64 pairs per GPU, and the headline configuration with the data-parallel
        # step's collectives running on RCCL in a one-rank group.  Neither touches `value`.
        import gc
        try:
            del model, batch
        except NameError:                  # (already released by the parity / cpu_baseline leg)
            pass
        gc.collect()
        torch.cuda.empty_cache()
        keep = ('value', 'unit', 'ms_per_step', 'steps', 'pairs_per_s', 'dtype', 'config', 'roofline', 'roofline_mfma',
                'hbm_peak_reserved_GB', 'last_loss', 'loss_scale', 'rccl_one_rank', 'rccl_version', 'dist_backend')
        out['configs4'] = _sub_bench(['--config', '4', '--pairs', '64', '--steps', '2', '--cfg4_parity', 'none'], 900, keep)
        out['rccl_one_rank'] = _sub_bench(['--rccl_one_rank', '--steps', '3'], 600, keep)
        if isinstance(out['rccl_one_rank'], dict) and 'last_loss' in out['rccl_one_rank']:
            out['rccl_one_rank']['last_loss_equals_headline'] = bool(out['rccl_one_rank']['last_loss'] == out['last_loss'])
    print(json.dumps(out))


if __name__ == '__main__':
    main()
