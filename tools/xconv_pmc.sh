#!/bin/bash
# SQ counters of the dense-convolution kernel at the decoder shape (one pass per counter group)
OUT=gpurun_out/${1:-xpmc}; mkdir -p $OUT; ROOT=$(pwd); export TMPDIR=/tmp MIOPEN_LOG_LEVEL=1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  ( cd /tmp && XCONV_NSHAPES=1 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/sq$i -o pmc -- \
      python $ROOT/tools/microbench_xconv.py nomiopen > $ROOT/$OUT/sq$i.log 2>&1 )
done
python tools/pmc_summary.py "$OUT/sq*/" 2>&1 | grep -E "xconv_kernel" > $OUT/sq_summary.txt
rm -rf $OUT/sq*/
cat $OUT/sq_summary.txt
