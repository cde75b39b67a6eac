#!/bin/bash
# A/B the warp+loss kernel on the GPU box: generations / pixels per thread-step / build variants, after the parity tests.
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
ROOT=$(pwd)
V=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants
run() { echo "== $*" >> $OUT/ab.log; env "$@" timeout 120 python tools/microbench_warp.py --iters 30 2>&1 | grep kernel >> $OUT/ab.log; }
timeout 900 python -m pytest tests/test_00_warp_loss_gpu.py -x -q > $OUT/pytest_warp.log 2>&1; tail -3 $OUT/pytest_warp.log
run DVD_X=default
for f in $V/libdvd_hip_*.so; do [ -f $f ] && run DVD_HIP_LIB=$f; done
run DVD_WARP_PX=4
run DVD_WARP_GEN=3
python - <<PY
import json
name=None
for l in open('$OUT/ab.log'):
    if l.startswith('=='): name=l.strip()
    else:
        d=json.loads(l); print('%-60s %.1f us  %.0f GB/s' % (name[-60:], d['ms_per_call_incl_memset_and_reduce']*1e3, d['GBps']))
PY
