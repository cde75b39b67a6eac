#!/bin/bash
set -u
OUT=gpurun_out/r03m; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_04_bnrelu_gpu.py tests/test_06_xconv_gpu.py tests/test_30_full_step_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
XCONV_NMUL=3 XCONV_ONLY=0,1,2,3,4,6 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv.jsonl 2> $OUT/xconv.err
python - <<'PY'
import json
for l in open('gpurun_out/r03m/xconv.jsonl'):
    r=json.loads(l); print('  ',r['shape'],'wgrad %.3f ms %.0f TF'%(r.get('xconv_wgrad_ms',0),r.get('xconv_wgrad_tfs',0)))
PY
timeout 900 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.log 2> $OUT/bench.err
tail -1 $OUT/bench.log | cut -c1-220
