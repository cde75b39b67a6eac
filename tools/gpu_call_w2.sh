#!/bin/bash
set -u
OUT=gpurun_out/r02w; mkdir -p $OUT; ROOT=$(pwd)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for gen in ${GENS:-4}; do
( cd /tmp && DVD_WARP_GEN=$gen timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr$gen -o t -- python $ROOT/tools/microbench_warp.py ${MB_ARGS:-} > $ROOT/$OUT/tr$gen.log 2>&1 )
f=$(find $OUT/tr$gen -name '*kernel_stats.csv' | head -1); echo "gen $gen"; python -c "
import csv,sys
for i,r in enumerate(csv.DictReader(open('$f'))):
    if i<7: print('%-60s calls %4s avg %10.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))" 
rm -rf $OUT/tr$gen
done
