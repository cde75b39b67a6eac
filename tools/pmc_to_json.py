#!/usr/bin/env python
"""Turn the rocprofv3 --pmc summary of the warp+loss micro-benchmark (tools/gpu_visit.sh, stage
pmc; FETCH_SIZE and WRITE_SIZE collected in separate passes) into profiles/warp_loss_pmc.json,
which bench.py reports as roofline.traffic.

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in
KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B read request for wide coalesced loads
(FETCH_SIZE = TCC_EA0_RDREQ x 64 B), i.e. exactly half of the bytes, so reads are doubled;
WRITE_SIZE agrees with TCC_EA0_WRREQ x 64 B and is taken as is."""
import json
import re
import sys


def _source_digest():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'dynamic-video-depth_amd'))
    from dvd_hip import build
    return build.source_digest(build.WARP_UNITS)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r01_warp_loss_pmc_summary.txt'
    dst = sys.argv[2] if len(sys.argv) > 2 else 'profiles/warp_loss_pmc.json'
    per = {}
    for line in open(src):
        m = re.match(r'\S+\s+(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.e+]+)\s+\(n=(\d+), ([0-9.]+) us\)', line)
        if not m:
            continue
        k = m.group(1).strip()
        per.setdefault(k, {})[m.group(2)] = float(m.group(3))
        per[k]['avg_us'] = float(m.group(5))
    kernels, total = {}, 0.0
    for k, v in per.items():
        rd = v.get('FETCH_SIZE', 0.0) * 1024.0 * 2.0
        wr = v.get('WRITE_SIZE', 0.0) * 1024.0
        kernels[k] = {'read_bytes': rd, 'write_bytes': wr, 'avg_us': v['avg_us']}
        total += rd + wr
    out = {'workload': '48 pairs x 384 x 672 (one dvd_warp_loss_fused launch sequence)',
           'hbm_bytes_per_launch': total, 'algorithmic_bytes_per_launch': 48 * 384 * 672 * 52,
           'corrections': 'FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), KiB -> bytes', 'kernels': kernels,
           'source': src, 'collected': sys.argv[3] if len(sys.argv) > 3 else 'rocprofv3 --pmc passes',
           'source_digest': _source_digest()}
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('hbm_bytes_per_launch', 'algorithmic_bytes_per_launch')}))


if __name__ == '__main__':
    main()
