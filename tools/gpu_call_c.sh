#!/bin/bash
set -u
OUT=gpurun_out/r02c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
timeout 900 python -m pytest tests/test_06_xconv_gpu.py -q --timeout 300 > $OUT/pytest_xconv.log 2>&1; tail -5 $OUT/pytest_xconv.log
timeout 300 python tools/microbench_xconv.py ${MB_ARGS:-nomiopen} 2>/dev/null | tee $OUT/micro_xconv.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], 'fwd %.3f ms %.0f TF/s  dgrad %.3f ms %.0f TF/s  wgrad %.3f ms %.0f TF/s' % (d['xconv_fwd_ms'], d['xconv_fwd_tfs'], d['xconv_dgrad_ms'], d['xconv_dgrad_tfs'], d['xconv_wgrad_ms'], d['xconv_wgrad_tfs']))"
