#!/bin/bash
# bench lines off the headline configuration, final binary: frame gaps 2 / 4, hourglass depth net, host-fed batches,
# and the 2-rank self-launch smoke test (gloo, two ranks sharing the one GPU)
set -u
OUT=gpurun_out/r03u; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
run() { tag=$1; shift; timeout 900 python bench.py --steps 2 --warmup 1 --no_cpu_baseline "$@" > $OUT/bench_$tag.log 2> $OUT/bench_$tag.err; echo "$tag: $(tail -1 $OUT/bench_$tag.log | cut -c1-200)"; }
run gap2 --gap 2
run gap4 --gap 4
run hourglass --depth hourglass
run hostfeed --feed host
DVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --pairs 8 --steps 1 --warmup 1 --no_cpu_baseline > $OUT/bench_2rank_gloo.log 2> $OUT/bench_2rank_gloo.err
echo "2rank: $(tail -1 $OUT/bench_2rank_gloo.log | cut -c1-300)"
