#!/bin/bash
# round 3, visit a: parity tests (with measured values logged), xconv block-shape A/B, bench line with the parity leg,
# the 2-rank self-launch smoke, kernel trace of the bench command.
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
rm -f $OUT/parity.jsonl
DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for cfg in 0 1 4; do
  XCONV_CFG=$cfg XCONV_NMUL=3 XCONV_NO_WGRAD=1 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_cfg$cfg.jsonl 2> $OUT/xconv_cfg$cfg.err
done
for cfg in 2 3; do
  XCONV_CFG=$cfg XCONV_NMUL=3 XCONV_NO_WGRAD=1 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_cfg$cfg.jsonl 2> $OUT/xconv_cfg$cfg.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/xconv_cfg*.jsonl')):
    print(f)
    for l in open(f):
        r=json.loads(l); print('  ',r['shape'],'fwd %.3f ms %.0f TF  dgrad %.3f ms %.0f TF'%(r['xconv_fwd_ms'],r['xconv_fwd_tfs'],r['xconv_dgrad_ms'],r['xconv_dgrad_tfs']))
PY
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench.log 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-1500; tail -3 $OUT/bench.err
DVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --pairs 8 --steps 1 --warmup 1 --no_cpu_baseline > $OUT/bench_2rank_gloo.log 2> $OUT/bench_2rank_gloo.err
echo "2-rank exit $?" >> $OUT/bench_2rank_gloo.err
tail -1 $OUT/bench_2rank_gloo.log | cut -c1-600; tail -3 $OUT/bench_2rank_gloo.err
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
echo "trace exit $?" >> $OUT/trace.log
python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
rm -rf $OUT/trace
head -30 $OUT/trace_summary.txt
