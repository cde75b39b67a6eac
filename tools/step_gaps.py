#!/usr/bin/env python
"""Idle time of the GPU inside steady training steps, from a rocprofv3 --kernel-trace csv of bench.py: the kernels of the last
`--steps` steps (delimited by the fused Adam launches of the depth net) in start order, the gaps between the end of one and
the start of the next, and the largest gaps with the kernels on either side.
  tools/step_gaps.py <kernel_trace.csv> [top]"""
import csv
import sys
from collections import defaultdict

rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-70:])
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
# a step ends with the Adam launches; take the span between the last two groups of adam kernels
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
groups = []
for i in adam:
    if groups and rows[i][0] - rows[groups[-1][-1]][1] < 50e6:      # launches of one step lie within 50 ms of each other
        groups[-1].append(i)
    else:
        groups.append([i])
if len(groups) < 3:
    sys.exit('fewer than three optimiser steps in the trace')
lo, hi = groups[-2][-1] + 1, groups[-1][-1] + 1
step = rows[lo:hi]
span = step[-1][1] - rows[lo - 1][1]
busy_end = rows[lo - 1][1]
gaps, busy = [], 0
by_after = defaultdict(lambda: [0, 0])
for s, e, name in step:
    if s > busy_end:
        gaps.append((s - busy_end, prev, name))
        by_after[name][0] += s - busy_end
        by_after[name][1] += 1
    busy += max(0, e - max(s, busy_end))
    if e > busy_end:
        busy_end, prev = e, name
print('last step: %d kernels, span %.2f ms, GPU busy %.2f ms, idle %.2f ms in %d gaps' %
      (len(step), span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps)))
for lim in (5e3, 20e3, 100e3, 1e6):
    print('  gaps > %6.0f us: %5d, %.2f ms' % (lim / 1e3, sum(1 for g in gaps if g[0] > lim), sum(g[0] for g in gaps if g[0] > lim) / 1e6))
print('largest gaps (us): before <- after')
for g, a, b in sorted(gaps, reverse=True)[:top]:
    print('  %8.1f  %s  ->  %s' % (g / 1e3, a, b))
print('idle in front of (summed, ms):')
for name, (t, n) in sorted(by_after.items(), key=lambda kv: -kv[1][0])[:top]:
    print('  %7.2f ms in %5d gaps  %s' % (t / 1e6, n, name))
