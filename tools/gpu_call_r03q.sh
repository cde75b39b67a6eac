#!/bin/bash
# warp+loss: one-block-per-CU tile shapes (96x64 / 1024 threads, 96x48 / 768 threads) against the production 96x32 / 512
set -u
OUT=gpurun_out/r03q; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests/test_06_xconv_gpu.py -m gpu -q -k elementwise > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "element-wise" $OUT/pytest.log | head -12
for rep in 1 2; do
  for t in -1 5 6; do
    timeout 200 python tools/microbench_warp.py --tile $t 2>/dev/null | tail -1 >> $OUT/warp_tiles.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r03q/warp_tiles.jsonl'):
    r=json.loads(l); print('tile %2d  %.1f us  frac %.3f' % (r['tile'], r['ms_per_call_incl_memset_and_reduce']*1e3, r['frac_of_8TBps']))
PY
