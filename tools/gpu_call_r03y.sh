#!/bin/bash
# ReLU masks of BatchNorm+ReLU sites applied in the consumer's backward-data epilogue (conv._Site) vs DVD_AB=no_maskfuse
set -u
OUT=gpurun_out/r03y; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest tests/test_09_fused_joins_gpu.py tests/test_06_xconv_gpu.py tests/test_30_full_step_gpu.py tests/test_31_benchmark_size_parity_gpu.py tests/test_20_model_surface_gpu.py -m gpu -q -s > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head; grep -E "worst vs float64" $OUT/pytest.log | head -8
timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_alias.log 2> $OUT/bench_alias.err
echo alias; tail -1 $OUT/bench_alias.log | cut -c1-220
DVD_AB=no_maskfuse timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_nomaskfuse.log 2> $OUT/bench_nomaskfuse.err
echo nomaskfuse; tail -1 $OUT/bench_nomaskfuse.log | cut -c1-220
