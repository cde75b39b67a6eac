#!/bin/bash
export TMPDIR=/tmp MIOPEN_LOG_LEVEL=1 MIOPEN_FIND_MODE=FAST HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03af; mkdir -p $OUT
timeout 100 python -m pytest tests/test_09_fused_joins_gpu.py tests/test_06_xconv_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -3
