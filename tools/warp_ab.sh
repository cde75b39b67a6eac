#!/bin/bash
# A/B the warp+loss kernel on the GPU box: parity tests on the product library, then the micro-benchmark on the product library
# and on every library under dvd_hip/lib/variants/ (tools/build_variant.sh), per tile shape in $TILES (default: auto and the
# 384-thread blocks); SQ=1 adds one counter pass (VALU instructions, busy / wait cycles, clock) per library and tile.
#   bash tools/warp_ab.sh <tag>      ->  gpurun_out/<tag>/{pytest_warp.log, ab.log, ab.txt, sq_*.txt}
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
ROOT=$(pwd); export TMPDIR=/tmp
V=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants
TILES=${TILES:-"-1 4"}
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 900 python -m pytest tests/test_00_warp_loss_gpu.py -x -q > $OUT/pytest_warp.log 2>&1; tail -3 $OUT/pytest_warp.log | cut -c1-200
fi
: > $OUT/ab.log
run() { lib=$1; tile=$2; echo "== lib=$(basename ${lib:-product}) tile=$tile" >> $OUT/ab.log
        env ${lib:+DVD_HIP_LIB=$lib} timeout 120 python tools/microbench_warp.py --iters 30 --tile $tile --px ${PX:-0} 2>&1 | grep kernel >> $OUT/ab.log; }
sq() { lib=$1; tile=$2; name=$(basename ${lib:-product} .so)_t$tile
       ( cd /tmp && env ${lib:+DVD_HIP_LIB=$lib} timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE \
           --output-format csv -d $ROOT/$OUT/sq_$name -o pmc -- python $ROOT/tools/microbench_warp.py --iters 3 --tile $tile --px ${PX:-0} > $ROOT/$OUT/sq_$name.log 2>&1 )
       python tools/pmc_summary.py "$OUT/sq_$name/" 2>&1 | grep -E "warp_loss_tiled|warp_loss_strip|combine" > $OUT/sq_$name.txt; rm -rf $OUT/sq_$name; }
for t in $TILES; do run "" $t; [ -n "${SQ:-}" ] && sq "" $t; done
for f in $V/libdvd_hip_*.so; do
  [ -f $f ] || continue
  for t in $TILES; do run $f $t; [ -n "${SQ:-}" ] && sq $f $t; done
done
python - <<PY > $OUT/ab.txt
import json
name=None
for l in open('$OUT/ab.log'):
    if l.startswith('=='): name=l.strip()
    else:
        d=json.loads(l); print('%-50s %.1f us  %.0f GB/s  frac %.3f' % (name[3:], d['ms_per_call_incl_memset_and_reduce']*1e3, d['GBps'], d['frac_of_8TBps']))
PY
cat $OUT/ab.txt
for f in $OUT/sq_*.txt; do [ -f $f ] && { echo $f; cut -c30-200 $f | grep -E "INSTS_VALU|WAVE_CYCLES|GRBM|ACTIVE_INST_VALU|WAIT" | grep -E "tiled|strip"; }; done
