#!/bin/bash
# Same-box A/B of end-to-end bench settings (box-to-box variance is ~20 %, so variants are only compared inside one call).
# usage: tools/bench_ab.sh <tag> "ENV1=.. --flag .." "ENV2=.. --flag .." ...
OUT=gpurun_out/$1; shift; mkdir -p $OUT
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=""; flags=""
  for tok in $spec; do case $tok in *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
  env $envs timeout 400 python bench.py --steps 2 --warmup 1 --no_cpu_baseline $flags > $OUT/run$i.log 2>&1
  python - <<PY
import json
for l in open('$OUT/run$i.log'):
    if l.startswith('{'):
        d=json.loads(l); print('%-50s %.4f iters/s  %.0f ms/step  loss %.6f' % ('$spec', d['value'], d['ms_per_step'], d['last_loss']))
PY
done
