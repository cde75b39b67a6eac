#!/usr/bin/env python
"""Per-step kernel time of every matrix-kernel class from TWO rocprofv3 --kernel-trace --stats runs of bench.py that differ only
in --steps: (calls_B - calls_A) / (steps_B - steps_A) launches per steady step (graph-capture / warm-up passes cancel) times the
mean duration of a launch in the longer run (without its slowest launch: first launches pay a code load).
Joins on the kernel-name fragments of dvd_hip.ops.FLOP_CLASS_KERNELS and writes profiles/mfma_roofline.json, which bench.py
reads for `roofline_mfma.top_kernels` (the algorithmic work per class is counted live by bench.py itself).

    python tools/mfma_roofline.py --a 2:<kernel_stats.csv> --b 5:<kernel_stats.csv> --mode fp32 --collected "<tag>" [--out ...]
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))


def load(path):
    """name -> (calls, total ns, max ns)"""
    out = {}
    for r in csv.DictReader(open(path)):
        name = r.get('Name') or r.get('KernelName') or ''
        out[name] = (int(r.get('Calls') or 0), float(r.get('TotalDurationNs') or 0.0), float(r.get('MaxNs') or 0.0))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--a', required=True)
    ap.add_argument('--b', required=True)
    ap.add_argument('--mode', choices=('fp32', 'fp16'), default='fp32')
    ap.add_argument('--collected', default='rocprofv3 --kernel-trace --stats of bench.py at two step counts')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'mfma_roofline.json'))
    a = ap.parse_args()
    from dvd_hip.ops import BYTE_CLASS_KERNELS, FLOP_CLASS_KERNELS
    from dvd_hip import build
    (na, pa), (nb, pb) = a.a.split(':', 1), a.b.split(':', 1)
    na, nb = int(na), int(nb)
    ta, tb = load(pa), load(pb)
    # calls per steady step from the DIFFERENCE of the two runs (graph captures, warm-up passes and set-up steps cancel); the
    # average duration of a call from the longer run without its single slowest call -- a first launch of a kernel can take
    # 100+ ms (code load), which as a difference of totals would swamp three steps of real time
    per, calls = {}, {}
    for k, (cb, totb, maxb) in tb.items():
        ca = ta.get(k, (0, 0.0, 0.0))[0]
        if cb > 1 and cb > ca:
            per[k] = (cb - ca) / float(nb - na) * (totb - maxb) / (cb - 1)
            calls[k] = (cb - ca) / float(nb - na)
    total = sum(v for v in per.values() if v > 0)
    classes = []
    for cls, frags in FLOP_CLASS_KERNELS.items():
        names = sorted(k for k in per if any(f in k for f in frags))
        ns = sum(per[k] for k in names)
        if ns > 0:
            classes.append({'class': cls, 'kernels': [n.split('(')[0][:80] for n in names], 'ms_per_step': ns / 1e6,
                            'share': ns / total})
    classes.sort(key=lambda r: -r['ms_per_step'])
    # the memory-bound helper classes (bench.py's roofline_helpers joins their live byte counts on these times) and what
    # neither list claims: ATen / runtime kernels and the fused warp+loss sequence (which has its own roofline)
    claimed = set(k for k in per if any(f in k for fr in FLOP_CLASS_KERNELS.values() for f in fr))
    helpers = []
    for cls, frags in BYTE_CLASS_KERNELS.items():
        names = sorted(k for k in per if k not in claimed and any(f in k for f in frags))
        claimed.update(names)
        ns = sum(per[k] for k in names)
        if ns > 0:
            helpers.append({'class': cls, 'kernels': [n.split('(')[0][:80] for n in names], 'ms_per_step': ns / 1e6,
                            'launches_per_step': sum(calls[k] for k in names), 'share': ns / total})
    helpers.sort(key=lambda r: -r['ms_per_step'])
    other = sorted(((per[k], k) for k in per if k not in claimed and per[k] > 0), reverse=True)
    rec = json.load(open(a.out)) if os.path.exists(a.out) else {}
    rec[a.mode] = {'collected': a.collected, 'steps': [na, nb], 'kernel_ms_per_step_all': total / 1e6, 'classes': classes,
                   'matrix_share_of_kernel_time': sum(c['share'] for c in classes), 'helpers': helpers,
                   'other_ms_per_step': sum(v for v, _ in other) / 1e6,
                   'other_kernels': [{'kernel': k.split('(')[0][-70:], 'ms_per_step': v / 1e6, 'launches_per_step': calls[k]}
                                     for v, k in other[:12]],
                   'source_digest': build.source_digest(None)}
    json.dump(rec, open(a.out, 'w'), indent=1)
    for c in classes:
        print('%-16s %8.2f ms/step  %5.1f %%' % (c['class'], c['ms_per_step'], 100 * c['share']))
    for c in helpers:
        print('%-16s %8.2f ms/step  %5.1f %%  %6.0f launches' % (c['class'], c['ms_per_step'], 100 * c['share'], c['launches_per_step']))
    print('all kernels %.1f ms/step; matrix classes %.1f %%; helpers %.1f ms; unclaimed %.1f ms' % (
        total / 1e6, 100 * rec[a.mode]['matrix_share_of_kernel_time'], sum(c['ms_per_step'] for c in helpers),
        rec[a.mode]['other_ms_per_step']))


if __name__ == '__main__':
    main()
