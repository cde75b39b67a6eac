#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for ab in 0 8 16 24; do
  DVD_WARP_GEN=4 DVD_WARP4_MODE=1 DVD_WARP_ABLATE=$ab timeout 200 python tools/microbench_warp.py --flow_sigma 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('gen4 sigma10 ablate $ab', 'ms', round(d['ms_per_call_incl_memset_and_reduce'],4))"
done
