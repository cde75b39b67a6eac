#!/bin/bash
# round 3, visit k: wide 1x1 weight-gradient kernel, cheaper max|.| scalars: parity suite, rates, bench + kernel trace
set -u
OUT=gpurun_out/r03k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
rm -f $OUT/parity.jsonl
DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q -k "not benchmark_size" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
XCONV_NMUL=3 XCONV_ONLY=8,9,10,11 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_1x1.jsonl 2> $OUT/xconv.err
python - <<'PY'
import json
for l in open('gpurun_out/r03k/xconv_1x1.jsonl'):
    r=json.loads(l); print('  ',r['shape'],'fwd %.3f ms %.0f TF  dgrad %.3f ms %.0f TF  wgrad %.3f ms %.0f TF'%(r['xconv_fwd_ms'],r['xconv_fwd_tfs'],r['xconv_dgrad_ms'],r['xconv_dgrad_tfs'],r.get('xconv_wgrad_ms',0),r.get('xconv_wgrad_tfs',0)))
PY
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $ROOT/$OUT/trace -o bench -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline > $ROOT/$OUT/trace.log 2>&1 )
echo "trace exit $?" >> $OUT/trace.log
python tools/rocprof_summary.py "$OUT/trace/**/*.db" > $OUT/trace_summary.txt 2>> $OUT/trace.log
mkdir -p $OUT/trace_keep; find $OUT/trace -name '*stats*.csv' -exec cp {} $OUT/trace_keep/ \;
rm -rf $OUT/trace
head -24 $OUT/trace_summary.txt | cut -c1-150
grep -o '"value": [0-9.]*, "unit"[^,]*, "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": [0-9.]*' $OUT/trace.log | tail -1
