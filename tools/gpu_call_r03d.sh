#!/bin/bash
# round 3, visit d: the fp16-pair arithmetic in xconv / xwgrad: parity suite, rates, bench line.
set -u
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
rm -f $OUT/parity.jsonl
DVD_PARITY_LOG=$ROOT/$OUT/parity.jsonl timeout 1500 python -m pytest tests -m gpu -q -s -k "not benchmark_size" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|error" $OUT/pytest.log | tail -5
grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
for cfg in 0 1 5; do
  XCONV_CFG=$cfg XCONV_NMUL=3 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_cfg$cfg.jsonl 2> $OUT/xconv_cfg$cfg.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03d/xconv_cfg*.jsonl')):
    print(f)
    for l in open(f):
        r=json.loads(l); print('  ',r['shape'],'fwd %.3f ms %.0f TF  dgrad %.3f ms %.0f TF  wgrad %.3f ms %.0f TF'%(r['xconv_fwd_ms'],r['xconv_fwd_tfs'],r['xconv_dgrad_ms'],r['xconv_dgrad_tfs'],r.get('xconv_wgrad_ms',0),r.get('xconv_wgrad_tfs',0)))
PY
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench.log 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-1800; tail -3 $OUT/bench.err
