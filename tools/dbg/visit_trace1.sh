OUT=gpurun_out/r05h; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/tr -o b -- python $ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/tr.log 2>&1 )
f=$(find $OUT/tr -name '*kernel_stats.csv' | head -1); cp $f $OUT/kernel_stats_2.csv; rm -rf $OUT/tr
grep -E "wamax|zero_words|xconv_pack|amax_kernel" $OUT/kernel_stats_2.csv | cut -c1-200
tail -1 $OUT/tr.log | cut -c1-150
timeout 600 python -m pytest tests/test_06_xconv_gpu.py -x -q 2>&1 | tail -2
