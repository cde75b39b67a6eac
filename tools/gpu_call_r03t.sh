#!/bin/bash
# rolling fragment prefetch in the 1x1 weight-gradient kernel (xwgrad1b) and the MLP's dW kernel vs the previous kernels
set -u
OUT=gpurun_out/r03t; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
V=$(pwd)/dynamic-video-depth_amd/dvd_hip/lib/variants/libdvd_hip_prevwg.so
timeout 900 python -m pytest tests/test_02_sf_mlp_gpu.py tests/test_06_xconv_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
for tag in new prev; do
  lib=$([ $tag = new ] && echo "" || echo $V)
  DVD_HIP_LIB=$lib XCONV_NMUL=3 XCONV_ONLY=8,9,10,11 timeout 300 python tools/microbench_xconv.py nomiopen > $OUT/xconv_$tag.jsonl 2> $OUT/xconv_$tag.err
  echo $tag; python - $OUT/xconv_$tag.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r=json.loads(l); print('  ',r['shape'],'wgrad %.3f ms %.0f TF'%(r['xconv_wgrad_ms'],r['xconv_wgrad_tfs']))
PY
  DVD_HIP_LIB=$lib timeout 300 python tools/microbench_mlp.py 2>/dev/null | grep '^{' | tail -1 > $OUT/mlp_$tag.json
  python -c "
import json; r=json.load(open('$OUT/mlp_$tag.json')); print('   mlp fwd %.2f ms %.0f TF  dx %.2f ms %.0f TF  dw %.2f ms %.0f TF  repro %s' % (r['fwd_ms'], r['fwd_tfs'], r['dx_ms'], r['dx_tfs'], r['dw_ms'], r['dw_tfs'], r['dw_bitwise_reproducible']))"
done
for tag in new prev; do
  lib=$([ $tag = new ] && echo "" || echo $V)
  DVD_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench_$tag.log 2> $OUT/bench_$tag.err
  echo $tag; tail -1 $OUT/bench_$tag.log | cut -c1-220
done
