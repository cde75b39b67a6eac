#!/bin/bash
export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
for ab in "" "no_bnfuse" "no_xwgrad3" "no_xconv" "no_xwgrad"; do
  DVD_AB=$ab timeout 200 python tools/debug/ckpt_debug3.py 2>&1 | grep -v amdgpu.ids | tail -4
done
DBG_GRAPHS=0 timeout 200 python tools/debug/ckpt_debug3.py 2>&1 | grep -v amdgpu.ids | tail -4
