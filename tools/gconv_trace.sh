#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/microbench_gconv.py: per-kernel mean durations of the grouped-convolution kernels.
#   gpurun -- 'bash tools/gconv_trace.sh'   ->  gpurun_out/gconv_trace/*kernel_stats.csv (+ the gconv rows on stdout)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/gconv_trace
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o g -- \
    python $ROOT/tools/microbench_gconv.py > $OUT/log.txt 2>&1 )
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gconv' in r['Name']:
        print('%-100s calls %4s  mean %8.1f us' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3))
PY
