#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSV outputs: mean counter value per kernel."""
import collections
import csv
import glob
import sys


def main():
    pats = sys.argv[1:] or ['gpurun_out/pmc*/']
    for pat in pats:
        for d in sorted(glob.glob(pat)):
            for f in glob.glob(d.rstrip('/') + '/*counter_collection.csv'):
                agg, dur = collections.defaultdict(list), collections.defaultdict(list)
                for r in csv.DictReader(open(f)):
                    k = r['Kernel_Name']
                    if k.startswith('__amd') or 'at::native' in k or k.startswith('void at'):
                        continue
                    k = k.split('(')[0][:70]
                    agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
                    dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
                for (k, c), v in sorted(agg.items()):
                    print('%-28s %-72s %-26s %14.5g  (n=%d, %.1f us)' % (d.rstrip('/').split('/')[-1], k, c,
                                                                          sum(v) / len(v), len(v),
                                                                          sum(dur[k]) / len(dur[k])))


if __name__ == '__main__':
    main()
