#!/bin/bash
# scene-flow MLP kernels: parity tests + microbench
set -u
OUT=gpurun_out/r02m; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests/test_02_sf_mlp_gpu.py -q --timeout 300 -rf > $OUT/pytest_mlp.log 2>&1; grep -v "^$" $OUT/pytest_mlp.log | grep -i "assert\|error\|passed\|failed\|mismatch\|Max \|off by" | head -40
timeout 300 python tools/microbench_mlp.py 2>&1 | tail -3 | tee $OUT/micro_mlp.json
