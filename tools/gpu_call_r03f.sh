#!/bin/bash
set -u
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
rm -f $OUT/parity.jsonl
DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -5
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
timeout 300 python tools/microbench_mlp.py > $OUT/mlp.json 2> $OUT/mlp.err; tail -3 $OUT/mlp.json | cut -c1-600
DVD_HIP_LIB=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants/libdvd_hip_mlpocc4.so timeout 300 python tools/microbench_mlp.py > $OUT/mlp_occ4.json 2>> $OUT/mlp.err; tail -3 $OUT/mlp_occ4.json | cut -c1-600
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench.log 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-2500; tail -3 $OUT/bench.err
