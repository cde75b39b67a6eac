#!/usr/bin/env python
"""profiles/warp_loss_sq.json from a tools/warp_ab.sh SQ=1 counter pass (or tools/warp_pmc_sq.sh summary) of the warp+loss
micro-benchmark at 48 x 384 x 672: VALU wave instructions, wave cycles and the effective clock of the tile kernel -- what
bench.py states as `roofline.valu_issue`.   python tools/sq_to_json.py <sq summary.txt> "<collected tag>" """
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _source_digest():
    sys.path.insert(0, os.path.join(ROOT, 'dynamic-video-depth_amd'))
    from dvd_hip import build
    return build.source_digest(build.WARP_UNITS)


def main():
    src, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    vals, us = {}, None
    lines = open(src).read().split('\n')
    # the production kernel of the launch sequence: the strip kernel (round 6) where it ran, else the tile kernel
    kernel = 'warp_loss_strip_kernel' if any('warp_loss_strip_kernel' in l for l in lines) else 'warp_loss_tiled_kernel'
    for line in lines:
        if kernel not in line:
            continue
        m = re.search(r'(\w+)\s+([0-9.e+]+)\s+\(n=\d+, ([0-9.]+) us\)', line)
        if m:
            vals[m.group(1)] = float(m.group(2))
            us = float(m.group(3))
    out = {'workload': '48 pairs x 384 x 672 (tools/microbench_warp.py)', 'pixels': 48 * 384 * 672,
           'valu_wave_instructions': vals['SQ_INSTS_VALU'], 'wave_cycles_quad': vals.get('SQ_WAVE_CYCLES'),
           'wait_any_quad': vals.get('SQ_WAIT_ANY'), 'tile_kernel_us_under_counters': us,
           # effective clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time
           'clock_hz': vals['GRBM_GUI_ACTIVE'] / 8.0 / (us * 1e-6) if 'GRBM_GUI_ACTIVE' in vals else 2.1e9,
           'kernel': kernel, 'source': src, 'collected': tag, 'source_digest': _source_digest()}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'warp_loss_sq.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
