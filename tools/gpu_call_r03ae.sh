#!/bin/bash
export TMPDIR=/tmp MIOPEN_LOG_LEVEL=1
OUT=gpurun_out/r03ae; mkdir -p $OUT
D=$(pwd)/dynamic-video-depth_amd/dvd_hip/lib/variants
for v in cur nors prevw1; do
  lib=$([ $v = cur ] && echo "" || echo $D/libdvd_hip_$v.so)
  DVD_HIP_LIB=$lib XCONV_NMUL=3 XCONV_ONLY=8 timeout 60 python tools/microbench_xconv.py nomiopen 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$v', r['shape'], 'wgrad %.3f ms %.0f TF' % (r['xconv_wgrad_ms'], r['xconv_wgrad_tfs']))"
done
