export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/gaps; mkdir -p $OUT; ROOT=$(pwd)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_tr -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_extras > $ROOT/$OUT/tr.log 2>&1 )
f=$(find /tmp/gaps_tr -name '*kernel_trace.csv' | head -1); ls -la $f
python tools/step_gaps.py $f 30 > $OUT/step_gaps.txt 2>&1; cat $OUT/step_gaps.txt
tail -2 $OUT/tr.log | cut -c1-300
