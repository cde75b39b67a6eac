#!/bin/bash
set -u
OUT=gpurun_out/r03l; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_04_bnrelu_gpu.py tests/test_06_xconv_gpu.py tests/test_02_sf_mlp_gpu.py tests/test_30_full_step_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
rm -rf $OUT/trace
head -30 $OUT/trace_summary.txt | cut -c1-150
timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-220
