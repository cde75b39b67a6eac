#!/bin/bash
# Repeated, interleaved A/B of the warp+loss micro-benchmark: the product library and every library under
# dvd_hip/lib/variants/, $REPS (default 5) runs each in alternation, --iters 100; prints min / median per library (one run of the
# micro-benchmark moves by +-3 % on a shared box, which is more than most of the differences that are being looked for).
#   bash tools/warp_ab_repeat.sh <tag>   ->  gpurun_out/<tag>/ab_repeat.txt
OUT=gpurun_out/${1:-abr}; mkdir -p $OUT
ROOT=$(pwd); V=$ROOT/dynamic-video-depth_amd/dvd_hip/lib/variants
LIBS="product"; for f in $V/libdvd_hip_*.so; do [ -f $f ] && LIBS="$LIBS $f"; done
: > $OUT/ab_repeat.log
# SHAPES="0 3": also alternate over strip shapes (csrc/warp_strip.hip kStripShapes) inside the same loop
for r in $(seq 1 ${REPS:-5}); do
  for lib in $LIBS; do
    for sh in ${SHAPES:-0}; do
      if [ $lib = product ]; then e=""; else e="DVD_HIP_LIB=$lib"; fi
      echo "== $(basename $lib) shape $sh" >> $OUT/ab_repeat.log
      env $e timeout 120 python tools/microbench_warp.py --iters 100 --strip_shape $sh ${WARP_ARGS:-} 2>&1 | grep kernel >> $OUT/ab_repeat.log
    done
  done
done
python - <<PY | tee $OUT/ab_repeat.txt
import json, collections
d = collections.defaultdict(list); name = None
for l in open('$OUT/ab_repeat.log'):
    if l.startswith('=='): name = l[3:].strip()
    else: d[name].append(json.loads(l)['ms_per_call_incl_memset_and_reduce'] * 1e3)
for k, v in d.items():
    v = sorted(v)
    print('%-40s min %.1f us  median %.1f us  max %.1f us  (%d runs)  frac(min) %.3f' % (k, v[0], v[len(v) // 2], v[-1], len(v), 644.087808 / v[0] / 8.0))
PY
