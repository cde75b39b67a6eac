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
    from dvd_hip.ops import FLOP_CLASS_KERNELS
    (na, pa), (nb, pb) = a.a.split(':', 1), a.b.split(':', 1)
    na, nb = int(na), int(nb)
    ta, tb = load(pa), load(pb)
    # calls per steady step from the DIFFERENCE of the two runs (graph captures, warm-up passes and set-up steps cancel); the
    # average duration of a call from the longer run without its single slowest call -- a first launch of a kernel can take
    # 100+ ms (code load), which as a difference of totals would swamp three steps of real time
    per = {}
    for k, (cb, totb, maxb) in tb.items():
        ca = ta.get(k, (0, 0.0, 0.0))[0]
        if cb > 1 and cb > ca:
            per[k] = (cb - ca) / float(nb - na) * (totb - maxb) / (cb - 1)
    total = sum(v for v in per.values() if v > 0)
    classes = []
    for cls, frags in FLOP_CLASS_KERNELS.items():
        names = sorted(k for k in per if any(f in k for f in frags))
        ns = sum(per[k] for k in names)
        if ns > 0:
            classes.append({'class': cls, 'kernels': [n.split('(')[0][:80] for n in names], 'ms_per_step': ns / 1e6,
                            'share': ns / total})
    classes.sort(key=lambda r: -r['ms_per_step'])
    rec = json.load(open(a.out)) if os.path.exists(a.out) else {}
    rec[a.mode] = {'collected': a.collected, 'steps': [na, nb], 'kernel_ms_per_step_all': total / 1e6, 'classes': classes,
                   'matrix_share_of_kernel_time': sum(c['share'] for c in classes)}
    json.dump(rec, open(a.out, 'w'), indent=1)
    for c in classes:
        print('%-16s %8.2f ms/step  %5.1f %%' % (c['class'], c['ms_per_step'], 100 * c['share']))
    print('all kernels %.1f ms/step; matrix classes %.1f %%' % (total / 1e6, 100 * rec[a.mode]['matrix_share_of_kernel_time']))


if __name__ == '__main__':
    main()
