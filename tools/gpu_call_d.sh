#!/bin/bash
set -u
OUT=gpurun_out/r02d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
for d in 0 1 2 4 8 16 7 23 31; do
  echo "dbg $d"; DVD_XCONV_DBG=$d XCONV_NSHAPES=1 timeout 300 python tools/microbench_xconv.py nomiopen 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], 'fwd %.3f ms %.0f TF/s  dgrad %.3f ms' % (d['xconv_fwd_ms'], d['xconv_fwd_tfs'], d['xconv_dgrad_ms']))"
done 2>&1 | tee $OUT/ablate.txt
