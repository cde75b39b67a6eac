#!/bin/bash
# Kernel traces (rocprofv3 --kernel-trace --stats) of bench.py variants: TAG, and VARIANTS = "name:args;name:args"
set -u
OUT=gpurun_out/${TAG:-trace}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
IFS=';' read -ra VS <<< "${VARIANTS:-fp32:}"
for v in "${VS[@]}"; do
  name=${v%%:*}; args=${v#*:}
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace_$name -o bench -- \
      python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline $args > $ROOT/$OUT/trace_$name.log 2>&1 )
  python tools/rocprof_summary.py "$OUT/trace_$name/**/*.db" > $OUT/trace_${name}_summary.txt 2>> $OUT/trace_$name.log
  rm -rf $OUT/trace_$name
  tail -1 $OUT/trace_$name.log | cut -c1-200
  head -40 $OUT/trace_${name}_summary.txt | cut -c1-170
done
