#!/bin/bash
set -u
OUT=gpurun_out/r02e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest tests/test_30_full_step_gpu.py -q -k "graphs or replayed" -s > $OUT/pytest_graphs.log 2>&1; tail -5 $OUT/pytest_graphs.log | cut -c1-300; grep "eager-vs" $OUT/pytest_graphs.log
for g in ${GRAPH_MODES:-1}; do
timeout 900 python bench.py --steps 3 --warmup 2 --no_cpu_baseline --depth_graphs $g > $OUT/bench_g$g.log 2> $OUT/bench_g$g.err; echo "bench exit $?"
tail -1 $OUT/bench_g$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graphs', $g, d['value'], d['ms_per_step'], d['roofline']['frac'], d['last_loss'])"
done
