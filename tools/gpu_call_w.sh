#!/bin/bash
set -u
OUT=gpurun_out/r02w; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_00_warp_loss_gpu.py -q --timeout 600 > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-220
rm -f $OUT/micro.jsonl
for args in "" "--flow_sigma 10" "--flow_sigma 30" "--smooth_flow" "--smooth_flow --flow_sigma 10" "--smooth_flow --flow_sigma 30"; do
  timeout 200 python tools/microbench_warp.py $args 2>/dev/null | tee -a $OUT/micro.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['flow_sigma'], d['smooth_flow'], 'ms', round(d['ms_per_call_incl_memset_and_reduce'],4), 'frac', round(d['frac_of_8TBps'],3))"
done
