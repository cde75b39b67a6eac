#!/bin/bash
# xconv: rolling fragment prefetch (256-row block shapes) vs the variant built with -DDVD_XCONV_ROLL=0
set -u
OUT=gpurun_out/r03r; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
V=$(pwd)/dynamic-video-depth_amd/dvd_hip/lib/variants/libdvd_hip_noroll.so
timeout 600 python -m pytest tests/test_06_xconv_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
for tag in roll noroll; do
  lib=$([ $tag = roll ] && echo "" || echo $V)
  DVD_HIP_LIB=$lib XCONV_NMUL=3 XCONV_NO_WGRAD=1 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_$tag.jsonl 2> $OUT/xconv_$tag.err
  echo $tag; python - $OUT/xconv_$tag.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r=json.loads(l); print('  ',r['shape'],'fwd %.3f ms %.0f TF  dgrad %.3f ms %.0f TF'%(r['xconv_fwd_ms'],r['xconv_fwd_tfs'],r['xconv_dgrad_ms'],r['xconv_dgrad_tfs']))
PY
done
for tag in roll noroll; do
  lib=$([ $tag = roll ] && echo "" || echo $V)
  DVD_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_$tag.log 2> $OUT/bench_$tag.err
  echo $tag; tail -1 $OUT/bench_$tag.log | cut -c1-220
done
