#!/bin/bash
# round 3, visit c: is xconv power limited?  (a) zero-filled operands (same instruction stream, no data switching),
# (b) a timing-only build that issues three of the six partial products.
set -u
OUT=gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_LOG_LEVEL=1
ROOT=$(pwd)
V=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants/libdvd_hip_x3prod.so
XCONV_CFG=1 XCONV_NMUL=3 XCONV_NO_WGRAD=1 XCONV_ONLY=2,4,8,9 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/base.jsonl 2> $OUT/err.log
XCONV_ZERO_INPUT=1 XCONV_CFG=1 XCONV_NMUL=3 XCONV_NO_WGRAD=1 XCONV_ONLY=2,4,8,9 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/zero.jsonl 2>> $OUT/err.log
DVD_HIP_LIB=$V XCONV_CFG=1 XCONV_NMUL=3 XCONV_NO_WGRAD=1 XCONV_ONLY=2,4,8,9 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/prod3.jsonl 2>> $OUT/err.log
DVD_HIP_LIB=$V XCONV_CFG=5 XCONV_NMUL=3 XCONV_NO_WGRAD=1 XCONV_ONLY=2,4,8,9 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/prod3_cfg5.jsonl 2>> $OUT/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/*.jsonl')):
    print(f)
    for l in open(f):
        r=json.loads(l); print('  ',r['shape'],'fwd %.3f ms %.0f TF  dgrad %.3f ms %.0f TF'%(r['xconv_fwd_ms'],r['xconv_fwd_tfs'],r['xconv_dgrad_ms'],r['xconv_dgrad_tfs']))
PY
