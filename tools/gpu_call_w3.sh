#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for args in "--flow_sigma 10" "--flow_sigma 10 --calm_border 48" "--flow_sigma 6" "--flow_sigma 4.5"; do
  timeout 200 python tools/microbench_warp.py $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$args', 'ms', round(d['ms_per_call_incl_memset_and_reduce'],4))"
done
