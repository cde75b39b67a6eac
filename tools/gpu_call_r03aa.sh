#!/bin/bash
# SQ counters of the grouped (32 rows per block) xconv shape
set -u
OUT=gpurun_out/r03aa; mkdir -p $OUT; ROOT=$(pwd)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST MIOPEN_LOG_LEVEL=1
cat > /tmp/gx.py <<'PY'
import sys, torch
sys.path.insert(0, 'dynamic-video-depth_amd'); sys.path.insert(0, 'tools')
from dvd_hip import conv as C
N, Cc, H, W = 48, 1024, 24, 42
x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, 32, 3, 3, device='cuda') * 0.05
for _ in range(5):
    y = C._xconv(x, w, None, None, False, False, Cc // 32)
torch.cuda.synchronize()
PY
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  ( cd $ROOT && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/q$i -o pmc -- python /tmp/gx.py > $ROOT/$OUT/q$i.log 2>&1 )
done
python tools/pmc_summary.py "$OUT/q*/" 2>&1 | grep -E "xconv_kernel" > $OUT/gx_sq_summary.txt
rm -rf $OUT/q*/
cut -c1-200 $OUT/gx_sq_summary.txt
