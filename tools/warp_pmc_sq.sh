#!/bin/bash
# SQ-side counters of the warp+loss kernel (issue-bound vs waiting), one 8-counter pass per group.
OUT=gpurun_out/${1:-sq}; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp
LIBV=${2:-}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  ( cd /tmp && env $LIBV timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/sq$i -o pmc -- \
      python $ROOT/tools/microbench_warp.py --iters 3 > $ROOT/$OUT/sq$i.log 2>&1 )
done
python tools/pmc_summary.py "$OUT/sq*/" 2>&1 | grep -E "warp_loss_tiled|warp_loss_strip|combine" > $OUT/sq_summary.txt
rm -rf $OUT/sq*/
cat $OUT/sq_summary.txt
