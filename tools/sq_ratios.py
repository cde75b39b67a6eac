#!/usr/bin/env python
"""Ratios from a pmc_summary listing of SQ counters (tools/gpu_visit.sh stages xsq / msq): per kernel the VALU, LDS and load
instructions per MFMA, the matrix pipe's busy share, the effective shader clock (GRBM_GUI_ACTIVE / wall time) and the wave-cycle
shares.  Values are per-launch averages.  SQ_VALU_MFMA_BUSY_CYCLES is the sum over the 1024 SIMDs of their busy cycles
(checked: exactly 32 x SQ_INSTS_MFMA for v_mfma_f32_32x32x16_f16); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import re
import sys

for path in sys.argv[1:]:
    d = collections.defaultdict(dict)
    for l in open(path):
        m = re.match(r'\S+\s+void dvd::(.+?)\s+(SQ_\w+|GRBM_\w+)\s+([\d.e+]+)\s+\(n=(\d+), ([\d.]+) us\)', l)
        if m:
            d[m.group(1).split('(')[0]][m.group(2)] = (float(m.group(3)), int(m.group(4)), float(m.group(5)))
    print('#', path)
    for k, v in d.items():
        g = lambda n: v.get(n, (0, 0, 0))[0]
        n, us = list(v.values())[0][1], list(v.values())[0][2]
        wc, mf = max(g('SQ_WAVE_CYCLES'), 1), max(g('SQ_INSTS_MFMA'), 1)
        gui = g('GRBM_GUI_ACTIVE')
        print('%s  n=%d  %.0f us' % (k[:80], n, us))
        print('    per MFMA: %.1f VALU (MFMA included), %.2f LDS, %.2f loads | matrix pipe busy %.0f %% | clock %.2f GHz | '
              'wave cycles: VALU %.0f %%, waiting %.0f %% (LDS %.0f %%) | LDS bank conflicts %.0f %% of its active cycles' % (
                  g('SQ_INSTS_VALU') / mf, g('SQ_INSTS_LDS') / mf, g('SQ_INSTS_VMEM_RD') / mf,
                  100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / max(1024 * gui / 8, 1), gui / 8 / max(us, 1e-9) / 1e3,
                  100 * g('SQ_ACTIVE_INST_VALU') / wc, 100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_WAIT_INST_LDS') / wc,
                  100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1)))
