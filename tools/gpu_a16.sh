#!/bin/bash
# fp16-activation visit: the A16 kernel tests, the convolution micro-benchmark in both storages, a short bench in both.
set -u
OUT=gpurun_out/${TAG:-a16}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
STAGES=${STAGES:-"tests micro bench"}
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
  DVD_PARITY_LOG=$(pwd)/$OUT/parity.jsonl timeout 900 python -m pytest tests/test_10_act_fp16_gpu.py -q -x ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1
  tail -25 $OUT/pytest.log | cut -c1-300
fi
if has micro; then
  XCONV_NMUL=3 XCONV_FP16=1 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_fp16.jsonl 2> $OUT/xconv_fp16.err
  XCONV_NMUL=3 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_fp32.jsonl 2> $OUT/xconv_fp32.err
  python - $OUT/xconv_fp16.jsonl $OUT/xconv_fp32.jsonl <<'PY'
import json, sys
for f in sys.argv[1:]:
    print(f)
    for l in open(f):
        r = json.loads(l)
        print('  ', r['shape'], ' '.join('%s=%.3g' % (k, v) for k, v in r.items() if k.endswith('_tfs') or k.endswith('fwd_ms')))
PY
  tail -3 $OUT/xconv_fp16.err
fi
if has bench; then
  timeout 900 python bench.py --no_cpu_baseline --steps 3 --act_fp16 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err
  cut -c1-400 $OUT/bench_fp16.json; tail -5 $OUT/bench_fp16.err
fi
