#!/bin/bash
# Resource usage (VGPRs, spills, scratch, LDS, occupancy) of every kernel of one translation unit, one line per kernel:
#   tools/kres.sh sf_mlp [-D...]
UNIT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/dynamic-video-depth_amd/dvd_hip
EXTRA=""
case $UNIT in warp_loss|unproject|elementwise|surfaces|upsample|consistency) EXTRA="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$ROOT/include -I$PKG/csrc $EXTRA "$@" \
  -Rpass-analysis=kernel-resource-usage -c $PKG/csrc/$UNIT.hip -o /tmp/kres_$UNIT.o 2>&1 |
  python3 -c "
import re,sys
cur=None
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: print(cur)
        cur=t.split(':',1)[1].strip()[:60].ljust(60)
    else:
        k,v=t.split(':',1)
        if k.strip() in ('VGPRs','AGPRs','VGPRs Spill','SGPRs Spill','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]'):
            cur+=' %s=%s'%(k.strip().split(' [')[0].replace(' ',''),v.strip())
if cur: print(cur)
"
