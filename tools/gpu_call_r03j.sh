#!/bin/bash
# round 3, visit j: kernel trace of the bench command with the fp16-pair arithmetic everywhere; extra bench lines (gap 2, gap 4, hourglass)
set -u
OUT=gpurun_out/r03j; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
echo "trace exit $?" >> $OUT/trace.log
python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
rm -rf $OUT/trace
head -32 $OUT/trace_summary.txt | cut -c1-160
tail -1 $OUT/trace.log | cut -c1-300
for g in 2 4; do
  timeout 900 python bench.py --steps 2 --warmup 1 --gap $g > $OUT/bench_gap$g.log 2> $OUT/bench_gap$g.err; echo "gap $g exit $?"
  tail -1 $OUT/bench_gap$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','hbm_peak_reserved_GB','hbm_graph_pools_GB')}, d['roofline']['avg_launch_ms'])"
done
timeout 900 python bench.py --steps 2 --warmup 1 --depth hourglass > $OUT/bench_hourglass.log 2> $OUT/bench_hourglass.err; echo "hourglass exit $?"
tail -1 $OUT/bench_hourglass.log | cut -c1-400; tail -3 $OUT/bench_hourglass.err
