#!/bin/bash
# per-kernel durations of the warp+loss launch sequence (rocprofv3 --kernel-trace --stats of the micro-benchmark)
OUT=gpurun_out/${1:-wtrace}; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
( cd /tmp && env ${2:-X=1} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr -o wl -- python $ROOT/tools/microbench_warp.py --iters 20 ${WARP_ARGS:-} > $ROOT/$OUT/trace.log 2>&1 )
f=$(find $OUT/tr -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee $OUT/kernel_stats.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('warp_', 'combine_')):   # tile / strip kernel, slab combine, finish
        print('%-90s calls %5s  avg %8.1f us' % (n[:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
rm -rf $OUT/tr
