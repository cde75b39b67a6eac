#!/bin/bash
set -u
OUT=gpurun_out/r02b; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest tests/test_06_xconv_gpu.py -q --timeout 300 -s > $OUT/pytest_xconv.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_xconv.log
grep -E "passed|failed|Error|err |worst|off" $OUT/pytest_xconv.log | head -60
timeout 600 python tools/microbench_xconv.py > $OUT/micro_xconv.jsonl 2> $OUT/micro_xconv.err; tail -3 $OUT/micro_xconv.err
cat $OUT/micro_xconv.jsonl
