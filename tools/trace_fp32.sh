#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py at --steps 2 and --steps 5 (fp32 storage) -> gpurun_out/$TAG/kernel_stats_fp32_{2,5}.csv,
# the inputs of tools/mfma_roofline.py (which can then run without a GPU).
OUT=gpurun_out/${TAG:-r05z}; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
for n in 2 5; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr_$n -o b -- python $ROOT/bench.py --steps $n --warmup 1 --no_cpu_baseline > $ROOT/$OUT/tr_$n.log 2>&1 )
  f=$(find $OUT/tr_$n -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats_fp32_$n.csv; rm -rf $OUT/tr_$n
done
ls -la $OUT
