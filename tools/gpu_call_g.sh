#!/bin/bash
# grouped xconv: parity tests, full-step fixtures, eager kernel trace, graph bench
set -u
OUT=gpurun_out/r02g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests/test_06_xconv_gpu.py -q -x --timeout 300 > $OUT/pytest_xconv.log 2>&1; tail -5 $OUT/pytest_xconv.log
timeout 900 python -m pytest tests/test_30_full_step_gpu.py tests/test_20_model_surface_gpu.py -q -x --timeout 600 > $OUT/pytest_full.log 2>&1; tail -5 $OUT/pytest_full.log
bash tools/gpu_trace_eager.sh r02g
timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | tee $OUT/bench.json | cut -c1-300
