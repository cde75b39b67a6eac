#!/usr/bin/env python
"""Timeline of the warp+loss launch sequence from a rocprofv3 --kernel-trace csv: per kernel its average duration and the
average idle gap in front of it (end of the previous kernel of the stream -> its start), over the micro-benchmark's timed
iterations.  Usage: tools/warp_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
seen = defaultdict(int)
SKIP = 6        # warm-up launches of the micro-benchmark (first launch loads the code object; a host sync follows them)
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-60:]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    seen[name] += 1
    if prev_end is not None and any(k in name for k in ('warp_', 'combine_')) and seen[name] > SKIP:
        dur[name] += e - s
        gap[name] += s - prev_end
        n[name] += 1
    prev_end = e
tot = 0.0
for k in dur:
    print('%-62s n %4d  avg %7.1f us  gap in front %6.2f us' % (k, n[k], dur[k] / n[k] / 1e3, gap[k] / n[k] / 1e3))
    tot += (dur[k] + gap[k]) / n[k] / 1e3
print('sum of kernels + gaps: %.1f us per launch sequence' % tot)
