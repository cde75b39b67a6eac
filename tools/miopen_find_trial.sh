#!/bin/bash
# Experiment: MIOpen find (benchmark) in DYNAMIC_HYBRID mode with the user find-db kept in the tree.
OUT=gpurun_out/${1:-find}; mkdir -p $OUT/miopen_db
export MIOPEN_LOG_LEVEL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for dc in 16; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --depth_chunk $dc > $OUT/chunk$dc.log 2>&1
  python -c "
import json,sys
for l in open('$OUT/chunk$dc.log'):
    if l.startswith('{'): d=json.loads(l); print('depth_chunk $dc', d['value'], d['ms_per_step'])"
done
start=$(date +%s)
MIOPEN_FIND_MODE=${FIND_MODE:-5} MIOPEN_USER_DB_PATH=$(pwd)/$OUT/miopen_db DVD_CUDNN_BENCHMARK=1 \
  timeout ${TUNE_TIMEOUT:-420} python bench.py --steps 2 --warmup 1 --no_cpu_baseline > $OUT/find.log 2> $OUT/find.err
echo "find exit $? after $(( $(date +%s) - start )) s"
python -c "
import json,sys
for l in open('$OUT/find.log'):
    if l.startswith('{'): d=json.loads(l); print('find mode', d['value'], d['ms_per_step'])"
du -sh $OUT/miopen_db; ls -la $OUT/miopen_db | head
