#!/bin/bash
# xwgrad3: staging order staggered between the waves of a SIMD vs the variant built with -DDVD_W3_STAGGER=0
set -u
OUT=gpurun_out/r03s; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
V=$(pwd)/dynamic-video-depth_amd/dvd_hip/lib/variants/libdvd_hip_nostagger.so
timeout 600 python -m pytest tests/test_06_xconv_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
for tag in stag nostag; do
  lib=$([ $tag = stag ] && echo "" || echo $V)
  DVD_HIP_LIB=$lib XCONV_NMUL=3 XCONV_ONLY=0,1,2,3,4,5,6 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_$tag.jsonl 2> $OUT/xconv_$tag.err
  echo $tag; python - $OUT/xconv_$tag.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r=json.loads(l); print('  ',r['shape'],'wgrad %.3f ms %.0f TF'%(r['xconv_wgrad_ms'],r['xconv_wgrad_tfs']))
PY
done
for tag in stag nostag; do
  lib=$([ $tag = stag ] && echo "" || echo $V)
  DVD_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_$tag.log 2> $OUT/bench_$tag.err
  echo $tag; tail -1 $OUT/bench_$tag.log | cut -c1-220
done
