#!/bin/bash
set -u
OUT=gpurun_out/r03i; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
timeout 200 python tools/debug/ckpt_debug3.py 2>&1 | grep -v amdgpu.ids | tail -4
rm -f $OUT/parity.jsonl
DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -5
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
