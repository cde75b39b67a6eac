#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (kernel-trace) into per-kernel stats:
calls, total/avg/min/max duration (us), share of GPU time.  Used to produce
the text summaries committed under profiles/."""
import glob
import sqlite3
import sys


def summarise(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(c.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start)/1000.0, avg(d.end-d.start)/1000.0, "
        "min(d.end-d.start)/1000.0, max(d.end-d.start)/1000.0, max(s.arch_vgpr_count), max(s.sgpr_count), "
        "max(d.group_segment_size) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
        % (disp, sym)))
    tot = sum(r[2] for r in rows) or 1.0
    out = ['%-90s %7s %12s %10s %10s %10s %6s %5s %5s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us',
                                                               'max_us', 'pct', 'vgpr', 'sgpr', 'lds')]
    for r in rows:
        out.append('%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5s %5s %7s' % (r[0][:90], r[1], r[2], r[3], r[4],
                                                                                r[5], 100.0 * r[2] / tot, r[6], r[7],
                                                                                r[8]))
    return '\n'.join(out)


if __name__ == '__main__':
    for pat in sys.argv[1:]:
        for db in sorted(set(glob.glob(pat, recursive=True))):
            print('# ' + db)
            print(summarise(db))
