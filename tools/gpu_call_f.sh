#!/bin/bash
set -u
OUT=gpurun_out/r02f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest tests/test_07_preprocess_gpu.py tests/test_08_feeder_gpu.py tests/test_20_model_surface_gpu.py tests/test_06_xconv_gpu.py -q --timeout 600 > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log | cut -c1-250
for f in hbm host; do
timeout 900 python bench.py --steps 3 --warmup 2 --no_cpu_baseline --feed $f > $OUT/bench_$f.log 2> $OUT/bench_$f.err; echo "bench exit $?"
tail -1 $OUT/bench_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['last_loss'])"
done
