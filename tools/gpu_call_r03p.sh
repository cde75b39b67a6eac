#!/bin/bash
set -u
OUT=gpurun_out/r03p; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests/test_05_upsample_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
timeout 200 python tools/microbench_upsample.py > $OUT/upsample.jsonl 2> $OUT/upsample.err; cut -c1-200 $OUT/upsample.jsonl
