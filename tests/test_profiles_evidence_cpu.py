"""The counter arithmetic DESIGN.md 5.6 rests on, checked against the committed rocprofv3 summaries (no GPU needed):
SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of 32 cycles per v_mfma_f32_32x32x16_f16, so the matrix pipe's busy share
and the effective clock follow from three counters (tools/sq_ratios.py)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r'\S+\s+void dvd::(.+?)\s+(SQ_\w+|GRBM_\w+)\s+([\d.e+]+)\s+\(n=(\d+), ([\d.]+) us\)')


def _counters(name):
    d = collections.defaultdict(dict)
    for l in open(os.path.join(ROOT, 'profiles', name)):
        m = LINE.match(l)
        if m:
            d[m.group(1).split('(')[0]][m.group(2)] = (float(m.group(3)), float(m.group(5)))
    return d


def test_mfma_busy_cycles_are_32_per_instruction():
    for name in ('r04_xconv_sq_counters_fp32.txt', 'r04_xconv_sq_counters_fp16.txt', 'r04_mlp_sq_counters.txt'):
        for kernel, c in _counters(name).items():
            busy, insts = c['SQ_VALU_MFMA_BUSY_CYCLES'][0], c['SQ_INSTS_MFMA'][0]
            assert abs(busy / (32.0 * insts) - 1.0) < 0.01, (name, kernel, busy, insts)


def test_busy_share_and_clock_of_the_weight_gradient_kernels():
    after, before = _counters('r04_xconv_sq_counters_fp32.txt'), _counters('r04q_xconv_sq_counters_fp32_before.txt')

    def share_clock(c):
        gui, us = c['GRBM_GUI_ACTIVE']
        return c['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (1024 * gui / 8), gui / 8 / us / 1e3

    b, cb = share_clock(before['xwgrad3_kernel<false>'])
    a, ca = share_clock(after['xwgrad3_kernel<false, true>'])
    assert 0.40 < b < 0.46 and 0.60 < a < 0.68            # 43 % -> 64 % of the matrix pipe
    assert 2.0 < cb < 2.2 and 1.65 < ca < 1.85            # ... and the clock gives a part of it back (power cap)
    assert after['xwgrad3_kernel<false, true>']['GRBM_GUI_ACTIVE'][1] < 0.86 * before['xwgrad3_kernel<false>']['GRBM_GUI_ACTIVE'][1]


def test_sq_ratios_tool_reads_the_committed_files():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'sq_ratios.py'),
                          os.path.join(ROOT, 'profiles', 'r04_xconv_sq_counters_fp32.txt')], capture_output=True, text=True, check=True).stdout
    assert 'xwgrad3_kernel<false, true>' in out and 'matrix pipe busy 64 %' in out
