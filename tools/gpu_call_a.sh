#!/bin/bash
# round-2 call A: whole GPU suite (no -x), warp+loss micro-benchmark over flow statistics, short bench
set -u
OUT=gpurun_out/r02a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for args in "" "--flow_sigma 10" "--flow_sigma 30" "--smooth_flow" "--smooth_flow --flow_sigma 10" "--smooth_flow --flow_sigma 30"; do
  timeout 200 python tools/microbench_warp.py $args >> $OUT/micro_warp.jsonl 2>> $OUT/micro_warp.err
done
cat $OUT/micro_warp.jsonl
timeout 900 python bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-600; tail -3 $OUT/bench.err
