#!/bin/bash
# BatchNorm mask / channel-sum pass with several small planes per block; final full GPU test run
set -u
OUT=gpurun_out/r03ac; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
DVD_PARITY_LOG=$(pwd)/$OUT/parity.jsonl timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $(pwd -P)/x -o bench -- true > /dev/null 2>&1 ) ; rm -rf /tmp/x
ROOT=$(pwd)
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
    python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
rm -rf $OUT/trace
grep -E "bnrelu|amax_kernel" $OUT/trace_summary.txt | cut -c1-150
tail -1 $OUT/trace.log | cut -c1-200
