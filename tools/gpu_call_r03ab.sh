#!/bin/bash
# per-channel sums of a pre-masked site's gradient from the 1x1 weight-gradient kernel (no pass of the site's own) vs DVD_AB=no_rowsum
set -u
OUT=gpurun_out/r03ab; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest tests/test_09_fused_joins_gpu.py tests/test_06_xconv_gpu.py tests/test_30_full_step_gpu.py tests/test_31_benchmark_size_parity_gpu.py tests/test_20_model_surface_gpu.py -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; grep -E "worst vs float64" $OUT/pytest.log | head -8 | cut -c1-200
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_rowsum.log 2> $OUT/bench_rowsum.err
echo rowsum; tail -1 $OUT/bench_rowsum.log | cut -c1-220
DVD_AB=no_rowsum timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_norowsum.log 2> $OUT/bench_norowsum.err
echo norowsum; tail -1 $OUT/bench_norowsum.log | cut -c1-220
