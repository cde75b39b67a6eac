#!/usr/bin/env python
"""Compile one HIP source to gfx950 assembly and print per-kernel instruction mix
(VALU / SALU / DS / VMEM, SGPR-spill v_readlane/v_writelane, IEEE divides) and
register counts.  Usage: tools/isa_stats.py <file.hip> [name-filter] [extra hipcc flags...]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dynamic-video-depth_amd', 'dvd_hip', 'csrc')


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    extra = sys.argv[3:]
    out = '/tmp/isa_stats.s'
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-S', '--cuda-device-only',
           '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, src, '-o', out] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr)
        sys.exit(1)
    text = open(out).read()
    lines = text.split('\n')
    meta = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    for i, name in starts:
        if filt not in name:
            continue
        body = []
        for l in lines[i + 1:]:
            if 's_endpgm' in l:
                break
            body.append(l)
        ins = [l.strip().split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        c = collections.Counter()
        for k in ins:
            if k.startswith('v_'):
                c['valu'] += 1
            elif k.startswith('s_'):
                c['salu'] += 1
            elif k.startswith('ds_'):
                c['ds'] += 1
            elif k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                c['vmem'] += 1
        cc = collections.Counter(ins)
        print('%s\n   total %d %s  readlane %d writelane %d div_scale %d mfma %d scratch %d' % (
            name[:110], len(ins), dict(c), cc['v_readlane_b32'], cc['v_writelane_b32'], cc['v_div_scale_f32'],
            sum(v for k, v in cc.items() if 'mfma' in k), sum(v for k, v in cc.items() if k.startswith('scratch_'))))


if __name__ == '__main__':
    main()
