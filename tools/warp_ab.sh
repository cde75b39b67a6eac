#!/bin/bash
# A/B the warp+loss kernel variants on the GPU box: generations 1/2/3, forced tile shapes, build variants.
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
ROOT=$(pwd)
V=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants
run() { echo "== $*" >> $OUT/ab.log; env "$@" timeout 120 python tools/microbench_warp.py --iters 30 2>&1 | grep kernel >> $OUT/ab.log; }
timeout 600 python -m pytest tests/test_warp_loss_gpu.py -x -q > $OUT/pytest_warp.log 2>&1; tail -3 $OUT/pytest_warp.log
timeout 900 python -m pytest tests/test_full_step_gpu.py -x -q > $OUT/pytest_full.log 2>&1; tail -3 $OUT/pytest_full.log
run DVD_WARP_GEN=3
run DVD_WARP_GEN=2
run DVD_WARP_GEN=1
for f in $V/*.so; do run DVD_HIP_LIB=$f; done
run DVD_WARP_GEN=3 DVD_WARP_ABLATE=1
run DVD_WARP_GEN=3 DVD_WARP_ABLATE=2
run DVD_WARP_GEN=3 DVD_WARP_ABLATE=7
cat $OUT/ab.log | python -c "
import sys, json
name=None
for l in sys.stdin:
    if l.startswith('=='): name=l.strip()
    else:
        d=json.loads(l); print('%-100s %.1f us  %.0f GB/s' % (name[-100:], d['ms_per_call_incl_memset_and_reduce']*1e3, d['GBps']))
"
bash tools/warp_pmc_sq.sh ${1:-ab}_sq DVD_WARP_GEN=3 > /dev/null 2>&1
grep -E "INSTS_VALU |INSTS_SALU|WAVE_CYCLES|ACTIVE_INST_VALU|WAIT_ANY|WAIT_INST_ANY" gpurun_out/${1:-ab}_sq/sq_summary.txt | grep tiled | sed 's/  */ /g'
